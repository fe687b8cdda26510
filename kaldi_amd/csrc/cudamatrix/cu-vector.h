// Device vector.  Public interface of the reference's CuVector<Real> (cudamatrix/cu-vector.h:34-115);
// see cu-matrix.h for the deliberate differences (no host back-up, asynchronous, float only).
#ifndef KALDI_AMD_CUDAMATRIX_CU_VECTOR_H_
#define KALDI_AMD_CUDAMATRIX_CU_VECTOR_H_
#include "cudamatrix/cu-matrix.h"

namespace kaldi {

template <typename Real>
class CuVector {
  static_assert(std::is_same<Real, float>::value, "CuVector is implemented for float only");
  typedef CuVector<Real> ThisType;

 public:
  CuVector() : dim_(0), data_(NULL), owner_(true) {}
  explicit CuVector(size_t dim) : dim_(0), data_(NULL), owner_(true) { Resize(dim); }
  CuVector(ThisType &&o) noexcept : dim_(o.dim_), data_(o.data_), owner_(o.owner_) {
    o.dim_ = 0;
    o.data_ = NULL;
    o.owner_ = true;
  }
  ~CuVector() { Destroy(); }

  size_t Dim() const { return dim_; }
  const Real *Data() const { return data_; }
  Real *Data() { return data_; }

  /// no-op when the size is unchanged; otherwise reallocates and zero-fills
  ThisType &Resize(size_t dim) {
    if (dim_ == dim) return *this;
    Destroy();
    if (dim == 0) return *this;
    void *p = NULL;
    CU_SAFE_CALL(hipK_malloc(&p, dim * sizeof(Real)));
    data_ = static_cast<Real *>(p);
    dim_ = dim;
    SetZero();
    return *this;
  }
  void Destroy() {
    if (data_ && owner_) {
      CuDevice::Instantiate().Synchronize();
      CU_SAFE_CALL(hipK_free(data_));
    }
    data_ = NULL;
    dim_ = 0;
    owner_ = true;
  }
  /// Makes this vector a window onto `dim` floats owned elsewhere (its own memory, if any, is freed); a later
  /// Resize to another size lets go of the window and allocates.
  void AttachView(Real *data, size_t dim) {
    Destroy();
    data_ = data;
    dim_ = dim;
    owner_ = false;
  }
  bool IsView() const { return !owner_; }
  ThisType &CopyFromVec(const CuVector<Real> &src) {
    Resize(src.Dim());
    CU_SAFE_CALL(hipK_memcpy(S(), data_, src.data_, dim_ * sizeof(Real), 2));
    return *this;
  }
  ThisType &CopyFromVec(const Vector<Real> &src) {
    Resize(src.Dim());
    CU_SAFE_CALL(hipK_memcpy(S(), data_, src.Data(), dim_ * sizeof(Real), 0));
    return *this;
  }
  void CopyToVec(Vector<Real> *dst) const {
    if (static_cast<size_t>(dst->Dim()) != dim_) dst->Resize(dim_);
    CU_SAFE_CALL(hipK_memcpy(S(), dst->Data(), data_, dim_ * sizeof(Real), 1));
  }
  void Read(std::istream &is, bool binary) {
    Vector<Real> tmp;
    tmp.Read(is, binary);
    CopyFromVec(tmp);
  }
  void Write(std::ostream &os, bool binary) const {
    Vector<Real> tmp;
    CopyToVec(&tmp);
    tmp.Write(os, binary);
  }

  void SetZero() {
    if (data_) CU_SAFE_CALL(hipK_memset_async(S(), data_, 0, dim_ * sizeof(Real)));
  }
  void Set(Real v) { CuProfileScope p(__func__); CU_SAFE_CALL(hipF_set_const(S(), data_, v, D())); }
  void Add(Real v) { CuProfileScope p(__func__); CU_SAFE_CALL(hipF_add_const(S(), data_, v, D())); }
  void Scale(Real v) { CuProfileScope p(__func__); CU_SAFE_CALL(hipF_scale(S(), data_, v, D())); }
  void Power(Real pw) { CuProfileScope p(__func__); CU_SAFE_CALL(hipF_power(S(), data_, pw, D())); }
  void Round() { CuProfileScope p(__func__); CU_SAFE_CALL(hipF_round(S(), data_, D())); }
  void ApplyLog() { CuProfileScope p(__func__); CU_SAFE_CALL(hipF_apply_log(S(), data_, D())); }
  void ApplyExp() { CuProfileScope p(__func__); CU_SAFE_CALL(hipF_apply_exp(S(), data_, D())); }
  void ApplyFloor(Real v) { CuProfileScope p(__func__); CU_SAFE_CALL(hipF_apply_floor(S(), data_, v, D())); }
  void ApplyTruncate(Real lo, Real hi) { CuProfileScope p(__func__); CU_SAFE_CALL(hipF_apply_truncate(S(), data_, lo, hi, D())); }
  void InvertElements() { CuProfileScope p(__func__); CU_SAFE_CALL(hipF_invert_elements(S(), data_, D())); }
  /// this = alpha*vec + beta*this
  void AddVec(Real alpha, const CuVector<Real> &vec, Real beta = 1.0) {
    CuProfileScope p(__func__);
    KALDI_ASSERT(vec.Dim() == Dim());
    CU_SAFE_CALL(hipF_add_mat(S(), alpha, vec.data_, beta, data_, D()));
  }
  /// this = alpha * (sum over the rows of mat) + beta*this
  void AddRowSumMat(Real alpha, const CuMatrix<Real> &mat, Real beta = 1.0) {
    CuProfileScope p(__func__);
    KALDI_ASSERT(static_cast<size_t>(mat.NumCols()) == Dim());
    CU_SAFE_CALL(hipF_add_row_sum_mat(S(), alpha, mat.Data(), mat.Dim(), beta, data_));
  }
  /// this = alpha * (sum over the columns of mat) + beta*this
  void AddColSumMat(Real alpha, const CuMatrix<Real> &mat, Real beta = 1.0) {
    CuProfileScope p(__func__);
    KALDI_ASSERT(static_cast<size_t>(mat.NumRows()) == Dim());
    CU_SAFE_CALL(hipF_add_col_sum_mat(S(), alpha, mat.Data(), mat.Dim(), beta, data_));
  }
  void MulElements(const CuVector<Real> &vec) {
    CuProfileScope p(__func__);
    KALDI_ASSERT(vec.Dim() == Dim());
    CU_SAFE_CALL(hipF_mul_elements(S(), data_, vec.data_, D()));
  }
  void DivElements(const CuVector<Real> &vec) {
    CuProfileScope p(__func__);
    KALDI_ASSERT(vec.Dim() == Dim());
    CU_SAFE_CALL(hipF_div_elements(S(), data_, vec.data_, D()));
  }
  /// The host-side vector of the reference's class (cudamatrix/cu-vector.h:103-109): empty whenever the data live on
  /// the device, which here is always.
  const Vector<Real> &Vec() const {
    static const Vector<Real> none;
    return none;
  }

 private:
  static khStream_t S() { return CuDevice::Instantiate().Stream(); }
  ::MatrixDim D() const {
    ::MatrixDim d = {1, static_cast<int32_t>(dim_), static_cast<int32_t>(dim_)};
    return d;
  }
  size_t dim_;
  Real *data_;
  bool owner_;  // false: a window onto memory owned elsewhere (AttachView)
  CuVector(const ThisType &) = delete;
  ThisType &operator=(const ThisType &) = delete;
};

template <typename Real>
std::ostream &operator<<(std::ostream &out, const CuVector<Real> &vec) {
  Vector<Real> tmp;
  vec.CopyToVec(&tmp);
  tmp.Write(out, false);
  return out;
}

}  // namespace kaldi
#endif
