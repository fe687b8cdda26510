// Shared helpers of libkaldi_hip.so (device layer of the nnet1 hot path, gfx950 only).
#ifndef KALDI_AMD_CSRC_HIP_COMMON_H_
#define KALDI_AMD_CSRC_HIP_COMMON_H_

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "kaldi_hip.h"

#define KH_EXPORT extern "C" __attribute__((visibility("default")))

namespace kh {

// Scratch memory of the kernels launched on one stream (split-K slabs, per-row partials).  One
// per stream: work queued on two streams may run at the same time and must not share it.
struct Workspace {
  hipStream_t stream = nullptr;
  void *ptr = nullptr;
  size_t bytes = 0;
  unsigned *ticket = nullptr;  // arrival counter of "last workgroup finishes the job" kernels; zero between launches
};

// Process-wide state: one GPU, one default stream, a scratch workspace per stream in use.
struct Context {
  bool initialised = false;
  int device = -1;
  hipStream_t stream = nullptr;
  int num_cu = 0;
  static constexpr int kMaxWorkspaces = 8;
  Workspace workspaces[kMaxWorkspaces];
  int num_workspaces = 0;
  char last_msg[256] = {0};
};
Context &ctx();

// Stream to launch on: the caller's, or the library default when it passed NULL.
inline hipStream_t pick(khStream_t s) { return s ? (hipStream_t)s : ctx().stream; }

// The workspace of stream `st`, grown to at least `bytes` (blocking when it has to reallocate).
int ensure_workspace(hipStream_t st, size_t bytes, void **ptr);
// Looks up, for `device` (current), the device address of the static scratch of hipF_xent_eval_static.
int static_scratch_setup(int device);
// The arrival counter of stream `st` (device memory, zero whenever no kernel that uses it is queued).
int stream_ticket(hipStream_t st, unsigned **ticket);

// What the row-softmax finish of a GEMM needs to go on to Xent::EvalVec in the same kernel.
struct XentArgs {
  const int32_t *tgt;
  float *diff;
  int lddiff;
  float *log_post;   // may be NULL (library scratch)
  int32_t *max_id;   // may be NULL
  long long *counters;
  double *loss;
  int write_post;    // 0: the posteriors themselves are not wanted
};

inline int launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? KH_OK : (int)e;
}

inline bool aligned16(const void *p) { return (((uintptr_t)p) & 15u) == 0; }

// d describes a matrix that a kernel may index: reject anything a kernel would fault on.
inline bool dim_ok(MatrixDim d) { return d.rows >= 0 && d.cols >= 0 && d.stride >= d.cols; }

}  // namespace kh

#define KH_CHECK_INIT()                                      \
  do {                                                       \
    if (!kh::ctx().initialised) {                            \
      int st__ = hipK_init(-1);                              \
      if (st__ != KH_OK) return st__;                        \
    }                                                        \
  } while (0)

#define KH_REQUIRE(cond)                                                                   \
  do {                                                                                     \
    if (!(cond)) {                                                                         \
      snprintf(kh::ctx().last_msg, sizeof(kh::ctx().last_msg), "%s: requirement failed: %s", \
               __func__, #cond);                                                           \
      return KH_ERR_INVALID_ARG;                                                           \
    }                                                                                      \
  } while (0)

#define KH_HIP(call)                           \
  do {                                         \
    hipError_t e__ = (call);                   \
    if (e__ != hipSuccess) return (int)e__;    \
  } while (0)

// ---- device-side helpers -------------------------------------------------------------------
#ifdef __HIPCC__
namespace kh {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Block-wide reductions for blocks of up to 1024 threads (16 waves); `red` holds 16 floats.
__device__ __forceinline__ float block_sum(float v, float *red) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  v = wave_sum(v);
  __syncthreads();
  if (lane == 0) red[wid] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += red[i];  // fixed order: every thread gets the same bits
  return t;
}
__device__ __forceinline__ float block_max(float v, float *red) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  v = wave_max(v);
  __syncthreads();
  if (lane == 0) red[wid] = v;
  __syncthreads();
  float t = red[0];
  for (int i = 1; i < nw; ++i) t = fmaxf(t, red[i]);
  return t;
}

// a * b + c as ONE fused operation, written out.  Where the sum of two products is left to the compiler
// (x * y + a * b) it may fuse either product into the add, and which one it picks can differ between two
// instantiations of the same source: the momentum term of the update formulas is pinned with this so that every
// kernel that applies them gives the same bits.
__device__ __forceinline__ float mul_add(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// One row of Mse::Eval by one 256-thread workgroup, and the total by the workgroup that arrives last: diff = a - b
// (stored when asked for), the row's sum of squares into row_sq[r]; the last of the `nrows` row workgroups to arrive
// (a device-scope counter, back at zero afterwards) adds the rows up in double, in a fixed order, and adds half of
// that to loss[0].  Row sums travel write-through / past the L1 (device-scope relaxed atomics), so no fence is needed.
// Scratch: red[16] floats, sd[256] doubles, one unsigned, all in LDS.  Used by the stand-alone kernel and by the
// row workgroups that ride in the RBM weight launch: the same bits from both.
__device__ __forceinline__ void mse_row_and_total(const float *a_row, const float *b_row, float *diff_row, int cols,
                                                  float *row_sq, int r, int nrows, unsigned *ticket, double *loss,
                                                  float *red, double *sd, unsigned *arrived) {
  float acc = 0.f;
  for (int c = threadIdx.x; c < cols; c += 256) {
    const float df = a_row[c] - b_row[c];
    if (diff_row) diff_row[c] = df;
    acc += df * df;
  }
  const float t = block_sum(acc, red);
  if (threadIdx.x == 0) {
    __hip_atomic_store(row_sq + r, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the row's sum has reached the L2
    *arrived = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (*arrived != (unsigned)nrows - 1u) return;
  double tot = 0.0;
  for (int i = threadIdx.x; i < nrows; i += 256)
    tot += (double)__hip_atomic_load(row_sq + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  sd[threadIdx.x] = tot;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sd[threadIdx.x] += sd[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    loss[0] += 0.5 * sd[0];
    __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// Per-element HybridTaus generator (three Tausworthe steps xor one LCG step): restates the reference's
// src/cudamatrix/cu-randkernels.cu:44-71 so that equal states give equal draws.  Uniform in (0,1): the product is
// formed in double and rounded to float; values that round to 0 or 1 are rejected and redrawn.
__device__ __forceinline__ uint32_t taus_step(uint32_t &z, int s1, int s2, int s3, uint32_t m) {
  const uint32_t b = (((z << s1) ^ z) >> s2);
  return z = (((z & m) << s3) ^ b);
}
__device__ __forceinline__ uint32_t lcg_step(uint32_t &z) { return z = 1664525u * z + 1013904223u; }
__device__ __forceinline__ float hybrid_taus(uint32_t &z1, uint32_t &z2, uint32_t &z3, uint32_t &z4) {
  float v;
  do {
    const uint32_t bits = taus_step(z1, 13, 19, 12, 4294967294u) ^ taus_step(z2, 2, 25, 4, 4294967288u) ^
                          taus_step(z3, 3, 11, 17, 4294967280u) ^ lcg_step(z4);
    v = (float)(2.3283064365387e-10 * (double)bits);
  } while (!(v > 0.0f && v < 1.0f));
  return v;
}

// 1/(1+exp(-x)) in fp32.  expf() here is the correctly scaled library call (not the fast
// intrinsic) so results stay within a few ulp of the reference's double evaluation.
__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); }

}  // namespace kh
#endif  // __HIPCC__

#endif  // KALDI_AMD_CSRC_HIP_COMMON_H_
