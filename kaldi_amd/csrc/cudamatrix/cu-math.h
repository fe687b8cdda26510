// cu:: free functions.  Public interface of the reference's cudamatrix/cu-math.h:35-111, each a
// single launch of the matching entry point of libkaldi_hip.so, plus the fused forms the nnet
// layer uses on its hot path.
#ifndef KALDI_AMD_CUDAMATRIX_CU_MATH_H_
#define KALDI_AMD_CUDAMATRIX_CU_MATH_H_
#include "cudamatrix/cu-matrix.h"
#include "cudamatrix/cu-stlvector.h"
#include "cudamatrix/cu-vector.h"

namespace kaldi {
namespace cu {

namespace detail {
inline khStream_t S() { return CuDevice::Instantiate().Stream(); }
template <typename Real>
void SameDims(const CuMatrix<Real> &a, const CuMatrix<Real> &b) {
  KALDI_ASSERT(a.NumRows() == b.NumRows() && a.NumCols() == b.NumCols() && a.Stride() == b.Stride());
}
}  // namespace detail

/// Y = max(0, X)
template <typename Real>
void Relu(const CuMatrix<Real> &X, CuMatrix<Real> *Y) {
  CuProfileScope p(__func__);
  detail::SameDims(X, *Y);
  CU_SAFE_CALL(hipF_relu(detail::S(), Y->Data(), X.Data(), X.Dim()));
}
/// Eout = (Y > 0) ? Ein : 0
template <typename Real>
void DiffRelu(const CuMatrix<Real> &Ein, const CuMatrix<Real> &Y, CuMatrix<Real> *Eout) {
  CuProfileScope p(__func__);
  detail::SameDims(Ein, *Eout);
  detail::SameDims(Y, *Eout);
  CU_SAFE_CALL(hipF_diff_relu(detail::S(), Eout->Data(), Ein.Data(), Y.Data(), Eout->Dim()));
}
/// Y = X > 4 ? X : log(1 + e^X)
template <typename Real>
void SoftRelu(const CuMatrix<Real> &X, CuMatrix<Real> *Y) {
  CuProfileScope p(__func__);
  detail::SameDims(X, *Y);
  CU_SAFE_CALL(hipF_softrelu(detail::S(), Y->Data(), X.Data(), X.Dim()));
}
template <typename Real>
void DiffSoftRelu(const CuMatrix<Real> &Ein, const CuMatrix<Real> &X, CuMatrix<Real> *Eout) {
  CuProfileScope p(__func__);
  detail::SameDims(Ein, *Eout);
  detail::SameDims(X, *Eout);
  CU_SAFE_CALL(hipF_diff_softrelu(detail::S(), Eout->Data(), Ein.Data(), X.Data(), Eout->Dim()));
}
/// Y = 1/(1+e^{-X})
template <typename Real>
void Sigmoid(const CuMatrix<Real> &X, CuMatrix<Real> *Y) {
  CuProfileScope p(__func__);
  detail::SameDims(X, *Y);
  CU_SAFE_CALL(hipF_sigmoid(detail::S(), Y->Data(), X.Data(), X.Dim()));
}
/// Eout = Y(1-Y) .* Ein
template <typename Real>
void DiffSigmoid(const CuMatrix<Real> &Ein, const CuMatrix<Real> &Y, CuMatrix<Real> *Eout) {
  CuProfileScope p(__func__);
  detail::SameDims(Ein, *Eout);
  detail::SameDims(Y, *Eout);
  CU_SAFE_CALL(hipF_diff_sigmoid(detail::S(), Eout->Data(), Ein.Data(), Y.Data(), Eout->Dim()));
}
/// row softmax, one pass
template <typename Real>
void Softmax(const CuMatrix<Real> &X, CuMatrix<Real> *Y) {
  CuProfileScope p(__func__);
  detail::SameDims(X, *Y);
  CU_SAFE_CALL(hipF_softmax(detail::S(), Y->Data(), X.Data(), X.Dim()));
}
template <typename Real>
void RegularizeL1(CuMatrix<Real> *wei, CuMatrix<Real> *grad, Real l1, Real lr) {
  CuProfileScope p(__func__);
  detail::SameDims(*wei, *grad);
  CU_SAFE_CALL(hipF_regularize_l1(detail::S(), wei->Data(), grad->Data(), l1, lr, wei->Dim()));
}
/// id[r] = argmax of row r (first maximum)
template <typename Real>
void FindRowMaxId(const CuMatrix<Real> &mat, CuStlVector<int32> *id) {
  CuProfileScope p(__func__);
  id->Resize(mat.NumRows());
  CU_SAFE_CALL(hipF_find_row_max_id(detail::S(), mat.Data(), NULL, id->Data(), mat.Dim()));
}
/// log_post_tgt[r] = log(out[r,tgt[r]]); out[r,tgt[r]] -= 1
template <typename Real>
void DiffXent(const CuStlVector<int32> &tgt, CuMatrix<Real> *net_out_or_diff, CuVector<Real> *log_post_tgt) {
  CuProfileScope p(__func__);
  KALDI_ASSERT(static_cast<MatrixIndexT>(tgt.Dim()) == net_out_or_diff->NumRows());
  log_post_tgt->Resize(tgt.Dim());
  CU_SAFE_CALL(hipF_diff_xent(detail::S(), tgt.Data(), net_out_or_diff->Data(), log_post_tgt->Data(),
                              net_out_or_diff->Dim()));
}
/// tgt.Row(i) = src.Row(copy_from_idx[i]) for i < copy_from_idx.Dim()
template <typename Real>
void Randomize(const CuMatrix<Real> &src, const CuStlVector<int32> &copy_from_idx, CuMatrix<Real> *tgt) {
  CuProfileScope p(__func__);
  KALDI_ASSERT(src.NumCols() == tgt->NumCols() && src.NumRows() == tgt->NumRows());
  KALDI_ASSERT(static_cast<MatrixIndexT>(copy_from_idx.Dim()) <= tgt->NumRows());
  ::MatrixDim dt = tgt->Dim();
  dt.rows = copy_from_idx.Dim();
  CU_SAFE_CALL(hipF_randomize(detail::S(), tgt->Data(), src.Data(), copy_from_idx.Data(), dt, src.Dim()));
}
/// splice frames: tgt[r, k*D+c] = src[clamp(r+offsets[k]), c]
template <typename Real>
void Expand(const CuMatrix<Real> &src, const CuStlVector<int32> &frame_offsets, CuMatrix<Real> *tgt) {
  CuProfileScope p(__func__);
  KALDI_ASSERT(src.NumCols() * static_cast<MatrixIndexT>(frame_offsets.Dim()) == tgt->NumCols());
  KALDI_ASSERT(src.NumRows() == tgt->NumRows());
  CU_SAFE_CALL(hipF_expand(detail::S(), tgt->Data(), src.Data(), frame_offsets.Data(), tgt->Dim(), src.Dim()));
}
/// column gather: tgt[r,c] = src[r, copy_from_indices[c]]
template <typename Real>
void Copy(const CuMatrix<Real> &src, const CuStlVector<int32> &copy_from_indices, CuMatrix<Real> *tgt) {
  CuProfileScope p(__func__);
  KALDI_ASSERT(static_cast<MatrixIndexT>(copy_from_indices.Dim()) == tgt->NumCols());
  KALDI_ASSERT(src.NumRows() == tgt->NumRows());
  CU_SAFE_CALL(hipF_copy(detail::S(), tgt->Data(), src.Data(), copy_from_indices.Data(), tgt->Dim(), src.Dim()));
}

}  // namespace cu
}  // namespace kaldi
#endif
