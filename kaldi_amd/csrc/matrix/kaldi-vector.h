// Host vector: storage, Kaldi text/binary I/O ("FV"/"DV", " [ a b c ]\n") and the few
// operations the nnet1 tools use.  Interface subset of the reference's Vector<Real>
// (matrix/kaldi-vector.h); formats follow matrix/kaldi-vector.cc:846-983.
#ifndef KALDI_AMD_MATRIX_KALDI_VECTOR_H_
#define KALDI_AMD_MATRIX_KALDI_VECTOR_H_
#include <ctype.h>
#include <math.h>
#include <strings.h>

#include <vector>

#include "matrix/matrix-common.h"

namespace kaldi {

template <typename Real>
class Vector {
 public:
  Vector() {}
  explicit Vector(MatrixIndexT dim, MatrixResizeType = kSetZero) : data_(dim, Real(0)) {}
  template <typename Other>
  explicit Vector(const Vector<Other> &v) { CopyFromVec(v); }

  MatrixIndexT Dim() const { return static_cast<MatrixIndexT>(data_.size()); }
  Real *Data() { return data_.data(); }
  const Real *Data() const { return data_.data(); }
  Real &operator()(MatrixIndexT i) { return data_[i]; }
  Real operator()(MatrixIndexT i) const { return data_[i]; }

  void Resize(MatrixIndexT dim, MatrixResizeType t = kSetZero) {
    if (t == kCopyData) data_.resize(dim, Real(0));
    else data_.assign(dim, Real(0));
  }
  void SetZero() { data_.assign(data_.size(), Real(0)); }
  void Set(Real v) { data_.assign(data_.size(), v); }
  template <typename Other>
  void CopyFromVec(const Vector<Other> &v) {
    data_.resize(v.Dim());
    for (MatrixIndexT i = 0; i < v.Dim(); ++i) data_[i] = static_cast<Real>(v(i));
  }
  /// double accumulator, like the reference's Sum()
  Real Sum() const {
    double s = 0.0;
    for (Real x : data_) s += x;
    return static_cast<Real>(s);
  }
  void Scale(Real a) { for (Real &x : data_) x *= a; }
  void Add(Real c) { for (Real &x : data_) x += c; }
  void ApplyLog() {
    for (Real &x : data_) {
      if (x < 0.0) KALDI_ERR << "Trying to take log of a negative number.";
      x = log(x);
    }
  }
  void ApplyPow(Real p) { for (Real &x : data_) x = pow(x, p); }
  void AddVec(Real alpha, const Vector<Real> &v) {
    KALDI_ASSERT(v.Dim() == Dim());
    for (MatrixIndexT i = 0; i < Dim(); ++i) data_[i] += alpha * v(i);
  }
  Real Max() const {
    KALDI_ASSERT(!data_.empty());
    Real m = data_[0];
    for (Real x : data_) m = x > m ? x : m;
    return m;
  }

  void Write(std::ostream &os, bool binary) const {
    if (!os.good()) KALDI_ERR << "Failed to write vector to stream: stream not good";
    if (binary) {
      WriteToken(os, binary, sizeof(Real) == 4 ? "FV" : "DV");
      WriteBasicType(os, binary, static_cast<int32>(Dim()));
      os.write(reinterpret_cast<const char *>(Data()), sizeof(Real) * Dim());
    } else {
      os << " [ ";
      for (Real x : data_) os << x << " ";
      os << "]\n";
    }
    if (!os.good()) KALDI_ERR << "Failed to write vector to stream";
  }

  void Read(std::istream &is, bool binary) {
    if (binary) {
      std::string tok;
      ReadToken(is, binary, &tok);
      if (tok != "FV" && tok != "DV") KALDI_ERR << "Failed to read vector from stream: expected FV or DV, got " << tok;
      int32 n;
      ReadBasicType(is, binary, &n);
      if (n < 0) KALDI_ERR << "Failed to read vector from stream: negative size";
      if (!ReadBinaryFloats(is, static_cast<size_t>(n), tok == "FV" ? 4 : 8, &data_))
        KALDI_ERR << "Failed to read vector from stream: truncated (size = " << n << ")";
      return;
    }
    std::string s;
    is >> s;
    if (is.fail()) KALDI_ERR << "Failed to read vector from stream: EOF while trying to read vector.";
    if (s == "[]") { data_.clear(); return; }
    if (s != "[") KALDI_ERR << "Failed to read vector from stream: expected \"[\" but got " << s;
    std::vector<Real> vals;
    for (;;) {
      const int c = is.peek();
      if (c == '-' || (c >= '0' && c <= '9')) {  // (what starts a number in the reference, kaldi-vector.cc:905)
        Real r;
        is >> r;
        if (is.fail()) KALDI_ERR << "Failed to read vector from stream: failed to read number.";
        if (!isspace(is.peek()) && is.peek() != ']')
          KALDI_ERR << "Failed to read vector from stream: expected whitespace after number.";
        vals.push_back(r);
      } else if (c == ' ' || c == '\t') {
        is.get();
      } else if (c == ']') {
        is.get();
        if (is.peek() == '\r') is.get();
        if (is.peek() == '\n') is.get();
        data_.swap(vals);
        return;
      } else if (c == -1) {
        KALDI_ERR << "Failed to read vector from stream: EOF while reading vector data.";
      } else if (c == '\n' || c == '\r') {
        KALDI_ERR << "Failed to read vector from stream: newline found while reading vector (maybe it's a matrix?)";
      } else {
        is >> s;
        if (!strcasecmp(s.c_str(), "inf") || !strcasecmp(s.c_str(), "infinity"))
          vals.push_back(std::numeric_limits<Real>::infinity());
        else if (!strcasecmp(s.c_str(), "nan"))
          vals.push_back(std::numeric_limits<Real>::quiet_NaN());
        else
          KALDI_ERR << "Failed to read vector from stream: expecting numeric vector data, got " << s;
      }
    }
  }

 private:
  std::vector<Real> data_;
};

}  // namespace kaldi
#endif
