// Host matrix (row-major, stride == cols): storage, Kaldi text/binary I/O ("FM"/"DM",
// " [\n  r0 ...\n  r1 ... ]\n") and the few operations the nnet1 tools use.  Interface subset of
// the reference's Matrix<Real> (matrix/kaldi-matrix.h); formats follow
// matrix/kaldi-matrix.cc:1038-1265.
#ifndef KALDI_AMD_MATRIX_KALDI_MATRIX_H_
#define KALDI_AMD_MATRIX_KALDI_MATRIX_H_
#include <string.h>
#include <strings.h>

#include <algorithm>

#include <vector>

#include "matrix/kaldi-vector.h"

namespace kaldi {

template <typename Real>
class Matrix {
 public:
  Matrix() : rows_(0), cols_(0) {}
  Matrix(MatrixIndexT r, MatrixIndexT c, MatrixResizeType = kSetZero) : rows_(0), cols_(0) { Resize(r, c); }
  template <typename Other>
  explicit Matrix(const Matrix<Other> &m) : rows_(0), cols_(0) { CopyFromMat(m); }

  MatrixIndexT NumRows() const { return rows_; }
  MatrixIndexT NumCols() const { return cols_; }
  MatrixIndexT Stride() const { return cols_; }
  Real *Data() { return data_.data(); }
  const Real *Data() const { return data_.data(); }
  Real *RowData(MatrixIndexT r) { return data_.data() + static_cast<size_t>(r) * cols_; }
  const Real *RowData(MatrixIndexT r) const { return data_.data() + static_cast<size_t>(r) * cols_; }
  Real &operator()(MatrixIndexT r, MatrixIndexT c) { return data_[static_cast<size_t>(r) * cols_ + c]; }
  Real operator()(MatrixIndexT r, MatrixIndexT c) const { return data_[static_cast<size_t>(r) * cols_ + c]; }

  /// kSetZero: all zero afterwards; kUndefined: contents unspecified (for buffers about to be
  /// overwritten whole: no pass over the memory when the allocation is already large enough)
  void Resize(MatrixIndexT r, MatrixIndexT c, MatrixResizeType type = kSetZero) {
    KALDI_ASSERT(r >= 0 && c >= 0);
    if (r == 0 || c == 0) r = c = 0;
    rows_ = r;
    cols_ = c;
    if (type == kUndefined) data_.resize(static_cast<size_t>(r) * c);
    else data_.assign(static_cast<size_t>(r) * c, Real(0));
  }
  void SetZero() { data_.assign(data_.size(), Real(0)); }
  void SetUnit() {
    SetZero();
    for (MatrixIndexT i = 0; i < rows_ && i < cols_; ++i) (*this)(i, i) = Real(1);
  }
  template <typename Other>
  void CopyFromMat(const Matrix<Other> &m) {
    Resize(m.NumRows(), m.NumCols());
    for (MatrixIndexT r = 0; r < rows_; ++r)
      for (MatrixIndexT c = 0; c < cols_; ++c) (*this)(r, c) = static_cast<Real>(m(r, c));
  }
  void Transpose() {
    Matrix<Real> t(cols_, rows_);
    for (MatrixIndexT r = 0; r < rows_; ++r)
      for (MatrixIndexT c = 0; c < cols_; ++c) t(c, r) = (*this)(r, c);
    Swap(&t);
  }
  void Swap(Matrix<Real> *o) {
    data_.swap(o->data_);
    std::swap(rows_, o->rows_);
    std::swap(cols_, o->cols_);
  }

  void Write(std::ostream &os, bool binary) const { WriteData(os, binary, Data(), rows_, cols_, cols_); }
  /// Writes rows x cols values, row r starting at data + r * stride, in the format of Write
  /// (kaldi-matrix.cc:1213-1251): lets a caller that holds the values elsewhere (a pinned download buffer
  /// laid out like the device matrix) write them without a Matrix.
  static void WriteData(std::ostream &os, bool binary, const Real *data, MatrixIndexT rows, MatrixIndexT cols,
                        MatrixIndexT stride) {
    if (!os.good()) KALDI_ERR << "Failed to write matrix to stream: stream not good";
    if (binary) {
      WriteToken(os, binary, sizeof(Real) == 4 ? "FM" : "DM");
      WriteBasicType(os, binary, static_cast<int32>(rows));
      WriteBasicType(os, binary, static_cast<int32>(cols));
      if (rows == 0 || cols == 0) {
        // (header only)
      } else if (stride == cols) {
        os.write(reinterpret_cast<const char *>(data), sizeof(Real) * static_cast<size_t>(rows) * cols);
      } else {
        // Rows of a pitched image, gathered into blocks of about 1 MiB: a stream passes every write of 1 KiB or more
        // straight to the file, and one system call per row of a few KiB costs more than copying the row once more.
        const size_t row_bytes = sizeof(Real) * static_cast<size_t>(cols);
        const size_t rows_per = std::max<size_t>(1, (static_cast<size_t>(1) << 20) / std::max<size_t>(row_bytes, 1));
        if (rows_per == 1) {
          for (MatrixIndexT r = 0; r < rows; ++r)
            os.write(reinterpret_cast<const char *>(data + static_cast<size_t>(r) * stride), row_bytes);
        } else {
          std::vector<char> block(std::min<size_t>(rows_per, rows) * row_bytes);
          for (MatrixIndexT r0 = 0; r0 < rows; r0 += static_cast<MatrixIndexT>(rows_per)) {
            const size_t n = std::min<size_t>(rows_per, rows - r0);
            for (size_t r = 0; r < n; ++r)
              memcpy(&block[r * row_bytes], data + (static_cast<size_t>(r0) + r) * stride, row_bytes);
            os.write(&block[0], n * row_bytes);
          }
        }
      }
    } else if (cols == 0) {
      os << " [ ]\n";
    } else {
      os << " [";
      for (MatrixIndexT r = 0; r < rows; ++r) {
        os << "\n  ";
        const Real *row = data + static_cast<size_t>(r) * stride;
        for (MatrixIndexT c = 0; c < cols; ++c) os << row[c] << " ";
      }
      os << "]\n";
    }
    if (!os.good()) KALDI_ERR << "Failed to write matrix to stream";
  }

  /// The header of a binary matrix ("FM" | "DM", rows, cols: kaldi-matrix.cc:1043-1062): after it the stream
  /// stands at the first of rows * cols * elem_bytes payload bytes, which ReadBinaryPayload reads and a caller
  /// that only wanted the size may skip.
  static void ReadBinaryHeader(std::istream &is, std::string *tok, int32 *rows, int32 *cols, size_t *elem_bytes) {
    ReadToken(is, true, tok);
    if (*tok != "FM" && *tok != "DM") KALDI_ERR << "Failed to read matrix from stream: expected FM or DM, got " << *tok;
    ReadBasicType(is, true, rows);
    ReadBasicType(is, true, cols);
    if (*rows < 0 || *cols < 0) KALDI_ERR << "Failed to read matrix from stream: negative size";
    *elem_bytes = (*tok == "FM") ? 4 : 8;
  }
  void Read(std::istream &is, bool binary) {
    if (binary) {
      std::string tok;
      int32 r, c;
      size_t elem;
      ReadBinaryHeader(is, &tok, &r, &c, &elem);
      ReadBinaryPayload(is, tok, r, c);
      return;
    }
    ReadText(is);
  }
  void ReadBinaryPayload(std::istream &is, const std::string &tok, int32 r, int32 c) {
    rows_ = cols_ = 0;
    if (r < 0 || c < 0) {
      data_.clear();
      KALDI_ERR << "Failed to read matrix from stream: dimensions " << r << " x " << c;
    }
    if (!ReadBinaryFloats(is, static_cast<size_t>(r) * c, tok == "FM" ? 4 : 8, &data_))
      KALDI_ERR << "Failed to read matrix from stream (" << r << " x " << c << "): truncated";
    if (r != 0 && c != 0) {
      rows_ = r;
      cols_ = c;
    }
  }
  void ReadText(std::istream &is) {
    std::string s;
    is >> s;
    if (is.fail()) KALDI_ERR << "Failed to read matrix from stream: expected \"[\", got EOF";
    if (s == "[]") { Resize(0, 0); return; }
    if (s != "[") KALDI_ERR << "Failed to read matrix from stream: expected \"[\", got \"" << s << '"';
    std::vector<Real> vals;   // all rows back to back
    size_t row_len = 0, cur = 0, nrows = 0;
    auto end_row = [&]() {
      if (cur == 0) return;
      if (nrows == 0) row_len = cur;
      else if (cur != row_len)
        KALDI_ERR << "Failed to read matrix from stream: inconsistent #cols: " << row_len << " vs. " << cur
                  << " (processing row " << nrows << ")";
      ++nrows;
      cur = 0;
    };
    for (;;) {
      const int c = is.peek();
      if (c == -1) {
        KALDI_ERR << "Failed to read matrix from stream: got EOF while reading matrix data";
      } else if (c == ']') {
        is.get();
        if (is.peek() == '\r') is.get();
        if (is.peek() == '\n') is.get();
        end_row();
        rows_ = static_cast<MatrixIndexT>(nrows);
        cols_ = static_cast<MatrixIndexT>(nrows ? row_len : 0);
        data_.swap(vals);
        return;
      } else if (c == '\n' || c == ';') {
        is.get();
        end_row();
      } else if ((c >= '0' && c <= '9') || c == '-') {  // (what starts a number in the reference, kaldi-matrix.cc:1228)
        Real r;
        is >> r;
        if (is.fail()) KALDI_ERR << "Failed to read matrix from stream: stream failure/EOF while reading matrix data.";
        vals.push_back(r);
        ++cur;
      } else if (isspace(c)) {
        is.get();
      } else {
        is >> s;
        if (!strcasecmp(s.c_str(), "inf") || !strcasecmp(s.c_str(), "infinity"))
          vals.push_back(std::numeric_limits<Real>::infinity());
        else if (!strcasecmp(s.c_str(), "nan"))
          vals.push_back(std::numeric_limits<Real>::quiet_NaN());
        else
          KALDI_ERR << "Failed to read matrix from stream: expecting numeric matrix data, got " << s;
        ++cur;
      }
    }
  }

 private:
  MatrixIndexT rows_, cols_;
  std::vector<Real> data_;
};

}  // namespace kaldi
#endif
