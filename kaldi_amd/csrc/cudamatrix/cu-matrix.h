// Device matrix.  Public interface of the reference's CuMatrix<Real>
// (cudamatrix/cu-matrix.h:43-205) on top of libkaldi_hip.so.  Differences kept deliberately:
//   * no host back-up matrix (Mat()): there is no CPU mode;
//   * operations are asynchronous on the CuDevice stream; only host<->device copies block;
//   * objects are movable but not copyable (the reference left copying undefined, which would
//     double-free device memory);
//   * only Real = float is supported (BaseFloat; nothing on the nnet1 path uses double).
#ifndef KALDI_AMD_CUDAMATRIX_CU_MATRIX_H_
#define KALDI_AMD_CUDAMATRIX_CU_MATRIX_H_
#include <type_traits>

#include "cudamatrix/cu-device.h"
#include "matrix/matrix-lib.h"

namespace kaldi {

template <typename Real> class CuVector;

/// Accumulates the wall time of one wrapper call into the CuDevice profile when profiling is on
/// (and synchronises, so that the time is the kernel's); does nothing otherwise.
class CuProfileScope {
 public:
  explicit CuProfileScope(const char *key) : key_(key), on_(CuDevice::Instantiate().Profiling()) {}
  ~CuProfileScope() {
    if (on_) {
      CuDevice::Instantiate().Synchronize();
      CuDevice::Instantiate().AccuProfile(key_, tim_.Elapsed());
    }
  }

 private:
  const char *key_;
  bool on_;
  Timer tim_;
};

template <typename Real>
class CuMatrix {
  static_assert(std::is_same<Real, float>::value, "CuMatrix is implemented for float only");
  typedef CuMatrix<Real> ThisType;

 public:
  CuMatrix() : num_rows_(0), num_cols_(0), stride_(0), data_(NULL), owner_(true), capacity_(0) {}
  CuMatrix(MatrixIndexT rows, MatrixIndexT cols)
      : num_rows_(0), num_cols_(0), stride_(0), data_(NULL), owner_(true), capacity_(0) {
    Resize(rows, cols);
  }
  CuMatrix(ThisType &&o) noexcept
      : num_rows_(o.num_rows_), num_cols_(o.num_cols_), stride_(o.stride_), data_(o.data_), owner_(o.owner_),
        capacity_(o.capacity_) {
    o.num_rows_ = o.num_cols_ = o.stride_ = 0;
    o.data_ = NULL;
    o.owner_ = true;
    o.capacity_ = 0;
  }
  ThisType &operator=(ThisType &&o) noexcept {
    if (this != &o) {
      Destroy();
      Swap(&o);
    }
    return *this;
  }
  ~CuMatrix() { Destroy(); }

  MatrixIndexT NumRows() const { return num_rows_; }
  MatrixIndexT NumCols() const { return num_cols_; }
  MatrixIndexT Stride() const { return stride_; }
  ::MatrixDim Dim() const {
    ::MatrixDim d = {num_rows_, num_cols_, stride_};
    return d;
  }
  const Real *Data() const { return data_; }
  Real *Data() { return data_; }
  const Real *RowData(MatrixIndexT r) const {
    KALDI_ASSERT(r >= 0 && r < num_rows_);
    return data_ + static_cast<size_t>(r) * stride_;
  }
  Real *RowData(MatrixIndexT r) {
    KALDI_ASSERT(r >= 0 && r < num_rows_);
    return data_ + static_cast<size_t>(r) * stride_;
  }
  size_t SizeInBytes() const { return static_cast<size_t>(num_rows_) * stride_ * sizeof(Real); }
  MatrixIndexT RowSizeInBytes() const { return num_cols_ * sizeof(Real); }
  MatrixIndexT StrideSizeInBytes() const { return stride_ * sizeof(Real); }

  /// No-op when the shape is unchanged; otherwise the matrix gets the new shape (rows start
  /// 256-byte aligned) and is zero-filled, padding included.  An allocation that is large enough is
  /// kept (per-utterance buffers change shape all the time; cudaMalloc/cudaFree per Resize, as in
  /// cu-matrix-inl.h:96-130, would synchronise the device each time).
  ThisType &Resize(MatrixIndexT rows, MatrixIndexT cols) {
    if (owner_ && num_rows_ == rows && num_cols_ == cols) return *this;  // a view always lets go
    KALDI_ASSERT(rows >= 0 && cols >= 0);
    const size_t pitch = (static_cast<size_t>(cols) * sizeof(Real) + 255) & ~static_cast<size_t>(255);
    if (owner_ && data_ && rows > 0 && cols > 0 && pitch * rows <= capacity_) {
      num_rows_ = rows;
      num_cols_ = cols;
      stride_ = static_cast<MatrixIndexT>(pitch / sizeof(Real));
      SetZero();
      return *this;
    }
    Destroy();
    if (rows == 0 || cols == 0) return *this;
    void *p = NULL;
    int32_t stride = 0;
    CU_SAFE_CALL(hipK_malloc_pitch(&p, &stride, rows, cols, sizeof(Real)));
    KALDI_ASSERT(static_cast<size_t>(stride) * sizeof(Real) == pitch);
    data_ = static_cast<Real *>(p);
    num_rows_ = rows;
    num_cols_ = cols;
    stride_ = stride;
    capacity_ = pitch * rows;
    SetZero();
    return *this;
  }
  void Destroy() {
    if (data_ && owner_) {
      // pending kernels may still use the buffer
      CuDevice::Instantiate().Synchronize();
      CU_SAFE_CALL(hipK_free(data_));
    }
    data_ = NULL;
    owner_ = true;
    capacity_ = 0;
    num_rows_ = num_cols_ = stride_ = 0;
  }
  /// Makes this matrix a non-owning window onto rows [row0, row0+rows) of `src` (no copy; not in the
  /// reference).  It stays valid while `src` keeps its allocation; any Resize to a new shape, or
  /// CopyFromMat, lets go of the window and allocates as usual.
  void AttachRows(const ThisType &src, MatrixIndexT row0, MatrixIndexT rows) {
    KALDI_ASSERT(row0 >= 0 && rows >= 0 && row0 + rows <= src.num_rows_);
    AttachView(const_cast<Real *>(src.data_) + static_cast<size_t>(row0) * src.stride_, rows, src.num_cols_, src.stride_);
  }
  /// Non-owning window onto device memory laid out like a CuMatrix (row pitch `stride` elements).
  void AttachView(Real *data, MatrixIndexT rows, MatrixIndexT cols, MatrixIndexT stride) {
    KALDI_ASSERT(rows >= 0 && cols >= 0 && stride >= cols);
    Destroy();
    if (rows == 0 || cols == 0) return;
    data_ = data;
    num_rows_ = rows;
    num_cols_ = cols;
    stride_ = stride;
    owner_ = false;
  }
  bool IsView() const { return !owner_; }
  void Swap(ThisType *o) {
    std::swap(owner_, o->owner_);
    std::swap(capacity_, o->capacity_);
    std::swap(num_rows_, o->num_rows_);
    std::swap(num_cols_, o->num_cols_);
    std::swap(stride_, o->stride_);
    std::swap(data_, o->data_);
  }

  ThisType &CopyFromMat(const CuMatrix<Real> &src) {
    Resize(src.NumRows(), src.NumCols());
    Copy2D(data_, stride_, src.data_, src.stride_, num_cols_, num_rows_, 2);
    return *this;
  }
  ThisType &CopyFromMat(const Matrix<Real> &src) {
    Resize(src.NumRows(), src.NumCols());
    Copy2D(data_, stride_, src.Data(), src.Stride(), num_cols_, num_rows_, 0);
    return *this;
  }
  void CopyToMat(Matrix<Real> *dst) const {
    if (dst->NumRows() != num_rows_ || dst->NumCols() != num_cols_) dst->Resize(num_rows_, num_cols_, kUndefined);
    Copy2D(dst->Data(), dst->Stride(), data_, stride_, num_cols_, num_rows_, 1);
  }
  /// this <- src for two matrices of one shape, written where this already lives: a window onto memory owned
  /// elsewhere stays a window (CopyFromMat sizes its target, and sizing lets go of a window).  Not in the reference.
  void AssignInPlace(const CuMatrix<Real> &src) {
    KALDI_ASSERT(src.NumRows() == num_rows_ && src.NumCols() == num_cols_);
    if (num_rows_ == 0 || &src == this) return;
    Copy2D(data_, stride_, src.data_, src.stride_, num_cols_, num_rows_, 2);
  }
  /// this <- src[ro:ro+r, co:co+c]; this becomes r x c first (cudamatrix/cu-matrix-inl.h:168-188)
  ThisType &CopyFromMat(const CuMatrix<Real> &src, MatrixIndexT ro, MatrixIndexT r, MatrixIndexT co, MatrixIndexT c) {
    KALDI_ASSERT(ro >= 0 && co >= 0 && r >= 0 && c >= 0 && ro + r <= src.NumRows() && co + c <= src.NumCols());
    KALDI_ASSERT(&src != this);
    Resize(r, c);
    Copy2D(data_, stride_, src.data_ + static_cast<size_t>(ro) * src.stride_ + co, src.stride_, c, r, 2);
    return *this;
  }
  /// dst[ro:ro+rows, co:co+cols] <- this.  A destination too small for the window is resized to end exactly at
  /// the window's last row and column, which zeroes what it held (cudamatrix/cu-matrix-inl.h:242-246).
  void CopyToMat(CuMatrix<Real> *dst, MatrixIndexT ro, MatrixIndexT co) const {
    KALDI_ASSERT(ro >= 0 && co >= 0 && dst != this);
    if (dst->NumRows() < ro + num_rows_ || dst->NumCols() < co + num_cols_) dst->Resize(ro + num_rows_, co + num_cols_);
    Copy2D(dst->data_ + static_cast<size_t>(ro) * dst->stride_ + co, dst->stride_, data_, stride_, num_cols_, num_rows_, 2);
  }
  void CopyRowsFromMat(int32 r, const CuMatrix<Real> &src, int32 src_ro, int32 dst_ro) {
    KALDI_ASSERT(r >= 0 && r + src_ro <= src.NumRows() && r + dst_ro <= NumRows() && src_ro >= 0 && dst_ro >= 0);
    KALDI_ASSERT(src.NumCols() == NumCols());
    Copy2D(data_ + static_cast<size_t>(dst_ro) * stride_, stride_, src.data_ + static_cast<size_t>(src_ro) * src.stride_,
           src.stride_, num_cols_, r, 2);
  }

  /// rows [src_ro, src_ro + r) of a host matrix into rows [dst_ro, ...) of this one (blocking; not in the reference)
  void CopyRowsFromMat(int32 r, const Matrix<Real> &src, int32 src_ro, int32 dst_ro) {
    KALDI_ASSERT(r >= 0 && r + src_ro <= src.NumRows() && r + dst_ro <= NumRows() && src_ro >= 0 && dst_ro >= 0);
    KALDI_ASSERT(src.NumCols() == NumCols());
    if (r == 0) return;
    Copy2D(data_ + static_cast<size_t>(dst_ro) * stride_, stride_, src.RowData(src_ro), src.Stride(), num_cols_, r, 0);
  }

  void Read(std::istream &is, bool binary) {
    Matrix<Real> tmp;
    tmp.Read(is, binary);
    CopyFromMat(tmp);
  }
  void Write(std::ostream &os, bool binary) const {
    Matrix<Real> tmp;
    CopyToMat(&tmp);
    tmp.Write(os, binary);
  }

  void SetZero() {
    if (data_) CU_SAFE_CALL(hipK_memset_async(S(), data_, 0, SizeInBytes()));
  }
  void Set(Real value) { CuProfileScope p(__func__); CU_SAFE_CALL(hipF_set_const(S(), data_, value, Dim())); }
  void PartSet(Real value, MatrixIndexT ro, MatrixIndexT r, MatrixIndexT co, MatrixIndexT c) {
    CuProfileScope p(__func__);
    CU_SAFE_CALL(hipF_part_set_const(S(), data_, value, Dim(), ro, r, co, c));
  }
  void Binarize(Real thres) { CuProfileScope p(__func__); CU_SAFE_CALL(hipF_binarize(S(), data_, thres, Dim())); }
  void Add(Real value) { CuProfileScope p(__func__); CU_SAFE_CALL(hipF_add_const(S(), data_, value, Dim())); }
  void ApplyLog() { CuProfileScope p(__func__); CU_SAFE_CALL(hipF_apply_log(S(), data_, Dim())); }
  void ApplyExp() { CuProfileScope p(__func__); CU_SAFE_CALL(hipF_apply_exp(S(), data_, Dim())); }
  void Power(Real pw) { CuProfileScope p(__func__); CU_SAFE_CALL(hipF_power(S(), data_, pw, Dim())); }
  void Scale(Real value) { CuProfileScope p(__func__); CU_SAFE_CALL(hipF_scale(S(), data_, value, Dim())); }
  void InvertElements() { CuProfileScope p(__func__); CU_SAFE_CALL(hipF_invert_elements(S(), data_, Dim())); }
  void MulElements(const CuMatrix<Real> &A) {
    CuProfileScope p(__func__);
    SameShape(A);
    CU_SAFE_CALL(hipF_mul_elements(S(), data_, A.data_, Dim()));
  }
  void DivElements(const CuMatrix<Real> &A) {
    CuProfileScope p(__func__);
    SameShape(A);
    CU_SAFE_CALL(hipF_div_elements(S(), data_, A.data_, Dim()));
  }
  void MulColsVec(const CuVector<Real> &scale);
  void MulRowsVec(const CuVector<Real> &scale);
  void DivColsVec(const CuVector<Real> &div);
  void DivRowsVec(const CuVector<Real> &div);
  /// this = alpha*A + beta*this
  void AddMat(Real alpha, const CuMatrix<Real> &A, Real beta = 1.0) {
    CuProfileScope p(__func__);
    SameShape(A);
    CU_SAFE_CALL(hipF_add_mat(S(), alpha, A.data_, beta, data_, Dim()));
  }
  void AddVecToCols(Real alpha, const CuVector<Real> &col, Real beta = 1.0);
  void AddVecToRows(Real alpha, const CuVector<Real> &row, Real beta = 1.0);
  void AddVecToPartialRows(Real alpha, MatrixIndexT offset, const CuVector<Real> &row, Real beta = 1.0);
  void PartAddMat(Real alpha, const CuMatrix<Real> &A, MatrixIndexT ro, MatrixIndexT co, Real beta = 1.0) {
    CuProfileScope p(__func__);
    KALDI_ASSERT(ro >= 0 && co >= 0 && ro + A.NumRows() <= num_rows_ && co + A.NumCols() <= num_cols_);
    CU_SAFE_CALL(hipF_part_add_mat(S(), alpha, A.data_, A.Dim(), ro, co, beta, data_, Dim()));
  }
  /// this = alpha * op(A) * op(B) + beta * this
  void AddMatMat(Real alpha, const CuMatrix<Real> &A, MatrixTransposeType transA, const CuMatrix<Real> &B,
                 MatrixTransposeType transB, Real beta) {
    CuProfileScope p(__func__);
    const MatrixIndexT m = (transA == kTrans) ? A.NumCols() : A.NumRows();
    const MatrixIndexT k = (transA == kTrans) ? A.NumRows() : A.NumCols();
    const MatrixIndexT k2 = (transB == kTrans) ? B.NumCols() : B.NumRows();
    const MatrixIndexT n = (transB == kTrans) ? B.NumRows() : B.NumCols();
    KALDI_ASSERT(k == k2 && m == num_rows_ && n == num_cols_);
    CU_SAFE_CALL(hipF_gemm(S(), transA, transB, m, n, k, alpha, A.data_, A.stride_, B.data_, B.stride_, beta, data_, stride_));
  }
  void LogAddExpMat(const CuMatrix<Real> &A) {
    CuProfileScope p(__func__);
    SameShape(A);
    CU_SAFE_CALL(hipF_log_add_exp_mat(S(), A.data_, data_, Dim()));
  }
  void ApplyRowL2UpperBound(Real bound) {
    CuProfileScope p(__func__);
    CU_SAFE_CALL(hipF_apply_row_l2upperbound(S(), data_, bound, Dim()));
  }
  /// The host-side matrix of the reference's class (cudamatrix/cu-matrix.h:187-193), which holds the data there only
  /// when no GPU is in use and is empty otherwise.  The data of this class always live on the device: always empty,
  /// and read-only, so that code which means to fill it does not compile.
  const Matrix<Real> &Mat() const {
    static const Matrix<Real> none;
    return none;
  }

 private:
  static khStream_t S() { return CuDevice::Instantiate().Stream(); }
  void SameShape(const CuMatrix<Real> &A) const {
    KALDI_ASSERT(A.num_rows_ == num_rows_ && A.num_cols_ == num_cols_ && A.stride_ == stride_);
  }
  static void Copy2D(void *dst, MatrixIndexT dstride, const void *src, MatrixIndexT sstride, MatrixIndexT cols,
                     MatrixIndexT rows, int kind) {
    if (rows == 0 || cols == 0) return;
    CuProfileScope p(kind == 0 ? "CuMatrix::CopyFromMatH2D" : kind == 1 ? "CuMatrix::CopyToMatD2H" : "CuMatrix::CopyD2D");
    CU_SAFE_CALL(hipK_memcpy_2d(S(), dst, dstride * sizeof(Real), src, sstride * sizeof(Real), cols * sizeof(Real), rows, kind));
  }
  MatrixIndexT num_rows_, num_cols_, stride_;
  Real *data_;
  bool owner_;  // false: a window onto memory owned elsewhere (AttachView)
  size_t capacity_;  // bytes of the allocation this matrix owns (>= rows * stride)
  CuMatrix(const ThisType &) = delete;
  ThisType &operator=(const ThisType &) = delete;
};

template <typename Real>
std::ostream &operator<<(std::ostream &out, const CuMatrix<Real> &mat) {
  Matrix<Real> tmp;
  mat.CopyToMat(&tmp);
  tmp.Write(out, false);
  return out;
}

}  // namespace kaldi

#include "cudamatrix/cu-vector.h"

namespace kaldi {
template <typename Real>
void CuMatrix<Real>::MulColsVec(const CuVector<Real> &scale) {
  CuProfileScope p(__func__);
  KALDI_ASSERT(static_cast<MatrixIndexT>(scale.Dim()) == num_cols_);
  CU_SAFE_CALL(hipF_mul_cols_vec(S(), data_, scale.Data(), Dim()));
}
template <typename Real>
void CuMatrix<Real>::MulRowsVec(const CuVector<Real> &scale) {
  CuProfileScope p(__func__);
  KALDI_ASSERT(static_cast<MatrixIndexT>(scale.Dim()) == num_rows_);
  CU_SAFE_CALL(hipF_mul_rows_vec(S(), data_, scale.Data(), Dim()));
}
template <typename Real>
void CuMatrix<Real>::DivColsVec(const CuVector<Real> &div) {
  CuProfileScope p(__func__);
  KALDI_ASSERT(static_cast<MatrixIndexT>(div.Dim()) == num_cols_);
  CU_SAFE_CALL(hipF_div_cols_vec(S(), data_, div.Data(), Dim()));
}
template <typename Real>
void CuMatrix<Real>::DivRowsVec(const CuVector<Real> &div) {
  CuProfileScope p(__func__);
  KALDI_ASSERT(static_cast<MatrixIndexT>(div.Dim()) == num_rows_);
  CU_SAFE_CALL(hipF_div_rows_vec(S(), data_, div.Data(), Dim()));
}
template <typename Real>
void CuMatrix<Real>::AddVecToCols(Real alpha, const CuVector<Real> &col, Real beta) {
  CuProfileScope p(__func__);
  if (static_cast<MatrixIndexT>(col.Dim()) != num_rows_)
    KALDI_ERR << "Non matching dimensions: Rows:" << num_rows_ << " VectorDim:" << col.Dim();
  CU_SAFE_CALL(hipF_add_vec_to_cols(S(), alpha, col.Data(), beta, data_, Dim()));
}
template <typename Real>
void CuMatrix<Real>::AddVecToRows(Real alpha, const CuVector<Real> &row, Real beta) {
  CuProfileScope p(__func__);
  if (static_cast<MatrixIndexT>(row.Dim()) != num_cols_)
    KALDI_ERR << "Non matching dimensions: Cols:" << num_cols_ << " VectorDim:" << row.Dim();
  CU_SAFE_CALL(hipF_add_vec_to_rows(S(), alpha, row.Data(), beta, data_, Dim()));
}
template <typename Real>
void CuMatrix<Real>::AddVecToPartialRows(Real alpha, MatrixIndexT offset, const CuVector<Real> &row, Real beta) {
  CuProfileScope p(__func__);
  // the reference refuses a vector that reaches the last column too (">=", cudamatrix/cu-matrix-inl.h:793-795)
  if (offset < 0 || static_cast<MatrixIndexT>(row.Dim()) + offset >= num_cols_)
    KALDI_ERR << "Incorrect dimensions: Cols:" << num_cols_ << " VectorDim:" << row.Dim() << " offset:" << offset;
  CU_SAFE_CALL(hipF_add_vec_to_partial_rows(S(), alpha, offset, row.Data(), row.Dim(), beta, data_, Dim()));
}
}  // namespace kaldi
#endif
