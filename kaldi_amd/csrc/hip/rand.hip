// Per-element HybridTaus generator (kh::hybrid_taus, common.h), uniform, Box-Muller gaussian, and the two fused
// consumers used by RBM CD-1 sampling.  Restates src/cudamatrix/cu-randkernels.cu:44-120 so that equal seeds give
// equal draws.
#include "common.h"

namespace {

using kh::hybrid_taus;

__device__ __forceinline__ float box_muller(uint32_t &z1, uint32_t &z2, uint32_t &z3, uint32_t &z4) {
  const float two_pi = 6.283185307179586476925286766558f;
  const float u0 = hybrid_taus(z1, z2, z3, z4), u1 = hybrid_taus(z1, z2, z3, z4);
  const float r = (float)sqrt(-2.0 * (double)logf(u0));
  return r * sinf(two_pi * u1);
}

enum { kUniform = 0, kGauss = 1, kBinarize = 2, kAddGauss = 3 };

// state arrays are indexed like the matrix (r*stride + c); 16 B of state + 4-8 B of data per element
template <int MODE>
__global__ __launch_bounds__(256) void rand_kernel(float *out, const float *in, float scale, uint32_t *z1,
                                                   uint32_t *z2, uint32_t *z3, uint32_t *z4, MatrixDim d) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= d.cols) return;
  for (int r = blockIdx.y; r < d.rows; r += gridDim.y) {
    const size_t idx = (size_t)r * d.stride + c;
    uint32_t a = z1[idx], b = z2[idx], e = z3[idx], f = z4[idx];
    float v;
    if (MODE == kUniform) {
      v = hybrid_taus(a, b, e, f);
    } else if (MODE == kGauss) {
      v = box_muller(a, b, e, f);
    } else if (MODE == kBinarize) {
      v = (in[idx] > hybrid_taus(a, b, e, f)) ? 1.0f : 0.0f;
    } else {
      v = scale * box_muller(a, b, e, f) + out[idx];
    }
    out[idx] = v;
    z1[idx] = a;
    z2[idx] = b;
    z3[idx] = e;
    z4[idx] = f;
  }
}

template <int MODE>
int launch_rand(khStream_t s, float *out, const float *in, float scale, uint32_t *z1, uint32_t *z2, uint32_t *z3,
                uint32_t *z4, MatrixDim d) {
  KH_CHECK_INIT();
  KH_REQUIRE(kh::dim_ok(d));
  if (d.rows == 0 || d.cols == 0) return KH_OK;
  KH_REQUIRE(out && z1 && z2 && z3 && z4 && (MODE != kBinarize || in != nullptr));
  int gx = (d.cols + 255) / 256, gy = d.rows;
  const int cap = 8192 / gx;
  if (gy > cap) gy = cap < 1 ? 1 : cap;
  hipLaunchKernelGGL((rand_kernel<MODE>), dim3(gx, gy), dim3(256), 0, kh::pick(s), out, in, scale, z1, z2, z3, z4,
                     d);
  return kh::launch_status();
}

}  // namespace

KH_EXPORT int hipF_rand(khStream_t s, float *mat, uint32_t *z1, uint32_t *z2, uint32_t *z3, uint32_t *z4,
                        MatrixDim d) {
  return launch_rand<kUniform>(s, mat, nullptr, 0.f, z1, z2, z3, z4, d);
}
KH_EXPORT int hipF_gauss_rand(khStream_t s, float *mat, uint32_t *z1, uint32_t *z2, uint32_t *z3, uint32_t *z4,
                              MatrixDim d) {
  return launch_rand<kGauss>(s, mat, nullptr, 0.f, z1, z2, z3, z4, d);
}
KH_EXPORT int hipF_binarize_probs_rand(khStream_t s, float *states, const float *probs, uint32_t *z1, uint32_t *z2,
                                       uint32_t *z3, uint32_t *z4, MatrixDim d) {
  return launch_rand<kBinarize>(s, states, probs, 0.f, z1, z2, z3, z4, d);
}
KH_EXPORT int hipF_add_gauss_noise(khStream_t s, float *mat, float gscale, uint32_t *z1, uint32_t *z2, uint32_t *z3,
                                   uint32_t *z4, MatrixDim d) {
  return launch_rand<kAddGauss>(s, mat, nullptr, gscale, z1, z2, z3, z4, d);
}
