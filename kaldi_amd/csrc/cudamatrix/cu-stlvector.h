// Device vector of integers.  Public interface of the reference's CuStlVector<IntType>
// (cudamatrix/cu-stlvector.h:33-88); int32 only (labels, argmax ids, gather indices).
#ifndef KALDI_AMD_CUDAMATRIX_CU_STLVECTOR_H_
#define KALDI_AMD_CUDAMATRIX_CU_STLVECTOR_H_
#include <vector>

#include "cudamatrix/cu-device.h"

namespace kaldi {

template <typename IntType>
class CuStlVector {
  static_assert(sizeof(IntType) == 4, "CuStlVector is implemented for 32-bit integers only");
  typedef CuStlVector<IntType> ThisType;

 public:
  CuStlVector() : dim_(0), data_(NULL), owner_(true) {}
  explicit CuStlVector(size_t dim) : dim_(0), data_(NULL), owner_(true) { Resize(dim); }
  ~CuStlVector() { Destroy(); }

  size_t Dim() const { return dim_; }
  const IntType *Data() const { return data_; }
  IntType *Data() { return data_; }

  ThisType &Resize(size_t dim) {
    if (owner_ && dim_ == dim) return *this;
    Destroy();
    if (dim == 0) return *this;
    void *p = NULL;
    CU_SAFE_CALL(hipK_malloc(&p, dim * sizeof(IntType)));
    data_ = static_cast<IntType *>(p);
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
    owner_ = true;
    dim_ = 0;
  }
  /// Non-owning window onto `count` elements of device memory owned elsewhere (no copy; not in the
  /// reference); Resize to another size lets go of it.
  void AttachView(IntType *data, size_t count) {
    Destroy();
    if (count == 0) return;
    data_ = data;
    dim_ = count;
    owner_ = false;
  }
  void AttachRange(const ThisType &src, size_t offset, size_t count) {
    KALDI_ASSERT(offset + count <= src.dim_);
    AttachView(const_cast<IntType *>(src.data_) + offset, count);
  }
  ThisType &CopyFromVec(const std::vector<IntType> &src) {
    Resize(src.size());
    if (dim_) CU_SAFE_CALL(hipK_memcpy(S(), data_, src.data(), dim_ * sizeof(IntType), 0));
    return *this;
  }
  /// device-to-device copy of `count` elements of src starting at src_offset
  void CopyFromCuVec(const CuStlVector<IntType> &src, size_t src_offset, size_t count) {
    KALDI_ASSERT(src_offset + count <= src.Dim());
    Resize(count);
    if (count) CU_SAFE_CALL(hipK_memcpy(S(), data_, src.data_ + src_offset, count * sizeof(IntType), 2));
  }
  void CopyToVec(std::vector<IntType> *dst) const {
    dst->resize(dim_);
    if (dim_) CU_SAFE_CALL(hipK_memcpy(S(), dst->data(), data_, dim_ * sizeof(IntType), 1));
  }
  void SetZero() {
    if (data_) CU_SAFE_CALL(hipK_memset_async(S(), data_, 0, dim_ * sizeof(IntType)));
  }
  void Set(IntType value) {
    ::MatrixDim d = {1, static_cast<int32_t>(dim_), static_cast<int32_t>(dim_)};
    CU_SAFE_CALL(hipI32_set_const(S(), reinterpret_cast<int32_t *>(data_), static_cast<int32_t>(value), d));
  }

 private:
  static khStream_t S() { return CuDevice::Instantiate().Stream(); }
  size_t dim_;
  IntType *data_;
  bool owner_;
  CuStlVector(const ThisType &) = delete;
  ThisType &operator=(const ThisType &) = delete;
};

template <typename IntType>
std::ostream &operator<<(std::ostream &out, const CuStlVector<IntType> &vec) {
  std::vector<IntType> tmp;
  vec.CopyToVec(&tmp);
  out << "[";
  for (size_t i = 0; i < tmp.size(); ++i) out << " " << tmp[i];
  out << " ]\n";
  return out;
}

}  // namespace kaldi
#endif
