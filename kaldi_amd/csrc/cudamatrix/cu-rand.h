// Device random numbers.  Public interface of the reference's CuRand<Real>
// (cudamatrix/cu-rand.h:31-76): one HybridTaus state per matrix element, seeded on demand from
// the host's rand() stream (RandInt(128, RAND_MAX)) whenever the matrix size changes, z1..z4 in
// that order (cu-rand-inl.h:34-79) -- so an unseeded process draws the reference's numbers.
#ifndef KALDI_AMD_CUDAMATRIX_CU_RAND_H_
#define KALDI_AMD_CUDAMATRIX_CU_RAND_H_
#include <vector>

#include "cudamatrix/cu-matrix.h"

namespace kaldi {

template <typename Real>
class CuRand {
 public:
  CuRand() : state_size_(0) {
    for (int i = 0; i < 4; ++i) z_[i] = NULL;
  }
  ~CuRand() { Free(); }

  /// (re)seeds the four state arrays for matrices of state_size = rows*stride elements
  void SeedGpu(MatrixIndexT state_size) {
    if (state_size != state_size_) {
      Free();
      for (int i = 0; i < 4; ++i) {
        void *p = NULL;
        CU_SAFE_CALL(hipK_malloc(&p, static_cast<size_t>(state_size) * sizeof(uint32)));
        z_[i] = static_cast<uint32 *>(p);
      }
      state_size_ = state_size;
    }
    std::vector<uint32> host(state_size);
    for (int i = 0; i < 4; ++i) {
      for (MatrixIndexT j = 0; j < state_size; ++j) host[j] = RandInt(128, RAND_MAX);
      CU_SAFE_CALL(hipK_memcpy(S(), z_[i], host.data(), host.size() * sizeof(uint32), 0));
    }
  }
  /// uniform (0,1)
  void RandUniform(CuMatrix<Real> *tgt) {
    CuProfileScope p(__func__);
    Prepare(*tgt);
    CU_SAFE_CALL(hipF_rand(S(), tgt->Data(), z_[0], z_[1], z_[2], z_[3], tgt->Dim()));
  }
  /// standard normal
  void RandGaussian(CuMatrix<Real> *tgt) {
    CuProfileScope p(__func__);
    Prepare(*tgt);
    CU_SAFE_CALL(hipF_gauss_rand(S(), tgt->Data(), z_[0], z_[1], z_[2], z_[3], tgt->Dim()));
  }
  /// states = (probs > uniform) ? 1 : 0, drawn and compared in one pass
  void BinarizeProbs(const CuMatrix<Real> &probs, CuMatrix<Real> *states) {
    CuProfileScope p(__func__);
    Prepare(probs);
    states->Resize(probs.NumRows(), probs.NumCols());
    CU_SAFE_CALL(hipF_binarize_probs_rand(S(), states->Data(), probs.Data(), z_[0], z_[1], z_[2], z_[3], probs.Dim()));
  }
  /// The four state arrays for matrices shaped like `m` (seeded first if the size has changed), for a kernel that
  /// draws inside another operation's launch; indexed like m (row * Stride() + column).
  void States(const CuMatrix<Real> &m, uint32 **z1, uint32 **z2, uint32 **z3, uint32 **z4) {
    Prepare(m);
    *z1 = z_[0]; *z2 = z_[1]; *z3 = z_[2]; *z4 = z_[3];
  }
  /// tgt += gscale * N(0,1), in one pass
  void AddGaussNoise(CuMatrix<Real> *tgt, Real gscale = 1.0) {
    CuProfileScope p(__func__);
    Prepare(*tgt);
    CU_SAFE_CALL(hipF_add_gauss_noise(S(), tgt->Data(), gscale, z_[0], z_[1], z_[2], z_[3], tgt->Dim()));
  }

 private:
  static khStream_t S() { return CuDevice::Instantiate().Stream(); }
  void Prepare(const CuMatrix<Real> &m) {
    const MatrixIndexT size = m.NumRows() * m.Stride();
    if (size != state_size_) SeedGpu(size);
  }
  void Free() {
    for (int i = 0; i < 4; ++i) {
      if (z_[i]) {
        CuDevice::Instantiate().Synchronize();
        CU_SAFE_CALL(hipK_free(z_[i]));
      }
      z_[i] = NULL;
    }
    state_size_ = 0;
  }
  uint32 *z_[4];
  MatrixIndexT state_size_;
  KALDI_DISALLOW_COPY_AND_ASSIGN(CuRand);
};

}  // namespace kaldi
#endif
