// Device side of the fp32 MFMA GEMM: parameter block, kernels and their launchers.  Included by gemm.hip (host
// logic: declarations only) and, with KHG_DEFINE_KERNELS, by the gemm_epi*.hip files, each of which compiles the
// kernels of one epilogue (the build runs them in parallel; one file with every instantiation took minutes).
#ifndef KALDI_AMD_CSRC_HIP_GEMM_KERNEL_H_
#define KALDI_AMD_CSRC_HIP_GEMM_KERNEL_H_
#include <type_traits>
#include <utility>

#include "common.h"

#ifndef KH_UPDATE_STORE_AUX
#define KH_UPDATE_STORE_AUX 16  // cache policy of the update epilogue's stores: 16 = sc1 (write-through)
#endif
#ifndef KH_SCHED
#define KH_SCHED 3  // K-loop form: 0 group-by-group fragments, 2 whole-tile fragments, 3 whole-tile except in the update GEMMs
#endif

namespace khg {

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum Epi {
  EPI_AXPBY = 0,   // C = alpha*acc + beta*C
  EPI_SLAB = 1,    // slab[ks] = acc                     (split-K partial)
  EPI_BIAS_ACT = 2,// C = act(bias[col] + acc)
  EPI_DSIG = 3,    // C = acc * y * (1 - y)
  EPI_UPDATE = 4,  // corr = g*acc + mom*corr ; W = W - l2*W - lr*corr
  EPI_RBM = 5,     // dW = g*acc + dW - lrl2*W ; W += dW
  EPI_GRAD = 6,    // C = alpha*acc, the update product without the weight step (data-parallel gradient), 16-byte stores
  EPI_BIAS_SAMPLE = 7  // C = sigmoid(bias[col] + acc) and C2 = (C > uniform) ? 1 : 0 with one HybridTaus draw per element
};

struct GemmParams {
  const float *A, *B;
  int lda, ldb;
  int M, N, K;
  int splitk, k_per_split;
  int a_bytes, b_bytes;  // operand footprints when both are below 2 GiB (buffer addressing), else 0
  int tiles_m, tiles_n;
  int n_fastest;  // tile order inside an XCD's run of tiles: 0 = down the rows first, 1 = along the columns first
  // EPI_RBM only, nk1 > 0: TWO products in one pass over the output.  K-tiles [0, nk1) multiply A, B into a first
  // set of accumulators, K-tiles [nk1, 2 nk1) multiply A2, B2 (same shapes, leading dimensions and footprints)
  // into a second one; K counts both ranges.  The epilogue then applies both statistics of the CD-1 step:
  //   d = alpha * acc1 + beta * dW ; d = alpha2 * acc2 + d ; d = -l2 * W + d ; dW = d ; W += d
  const float *A2, *B2;
  float alpha2;
  int nk1;
  float alpha, beta;
  float *C;
  int ldc;
  size_t slab_stride;
  const float *bias;
  const float *y;
  int ldy;
  const float *W;  // weights read by the update epilogues ...
  int ldw;
  float *Wout;     // ... and where the new ones go (may alias W)
  int ldwout;
  float lr, l2, mom;
  int act;
  // EPI_BIAS_SAMPLE: the sampled states and the per-element generator state, all indexed like C (r * ldc + c)
  float *C2;
  unsigned *z1, *z2, *z3, *z4;
  int dyn_lds;     // extra LDS bytes to request at launch (caps how many workgroups share a compute unit)
  // EPI_UPDATE: footprints of W / Wout / C (< 2 GiB) when the epilogue may use 16-byte buffer accesses on them
  // (all 16-byte aligned, leading dimensions and N multiples of 4), else 0: see the epilogue
  int w_bytes, wout_bytes, c_bytes;
  // EPI_UPDATE only: the bias half of the update, `bias_tiles` groups of 64 outputs done inside the launch
  // (bias_io -= lr * (alpha * colsum(A) + mom * bcorr); A is the error matrix, K x M as stored)
  float *bias_io;
  float *bcorr;
  int bias_tiles;
  // EPI_RBM with two products, bias_tiles > 0: both bias statistics of the CD-1 step by workgroups at the end of the
  // launch, ceil(M / 64) for the hidden units (columns of A / A2; bias_io, bcorr = hidden bias and its momentum
  // term) and ceil(N / 64) for the visible ones (columns of B / B2; bias2_io, bcorr2)
  float *bias2_io;
  float *bcorr2;
  // ... and, mse_rows > 0, the reconstruction error of the step by one more workgroup per row behind those:
  // mse_loss[0] += 1/2 * sum((B - B2)^2) over the mse_rows x N elements (B = the reconstruction, B2 = the data)
  int mse_rows;
  float *mse_row_sq;     // mse_rows floats of scratch
  unsigned *mse_ticket;  // arrival counter, zero at rest
  double *mse_loss;
#ifdef KH_GEMM_STAMPS
  unsigned long long *stamps;
#endif
};

constexpr int kBK = 32;

// Several weight-update products in ONE launch (64 x 64 tiles, both operands row-major as stored: e^T * in).
// Problem i owns blocks [first[i], first[i + 1]); every first[i] is a multiple of 8, so a block keeps its residue
// modulo 8 (its XCD under round-robin placement) inside its problem and the XCD-aware tile decode keeps its meaning;
// blocks past a problem's tile count are padding and leave at once.
constexpr int kMaxGroup = 8;
struct GemmGroup {
  int n;
  int first[kMaxGroup + 1];
  GemmParams p[kMaxGroup];
};
void launch_update_group(hipStream_t st, const GemmGroup &g, bool big);  // big: 128 x 128 tiles, else 64 x 64
// the same launch with the EPI_GRAD epilogue; `done` (may be null) is an event bound to the launch's own completion
void launch_grad_group(hipStream_t st, const GemmGroup &g, bool big, hipEvent_t done);

// Dependent forward products in one launch (gemm_chain_kernel below).
constexpr int kMaxChain = 8;
struct GemmChain {
  int n;
  unsigned epoch;
  unsigned *counters;  // [kMaxChain][tiles_m]
  int *fail;
  GemmParams p[kMaxChain];
};
void launch_forward_chain(hipStream_t st, const GemmChain &g, int cfg);

// launches the main kernel for tile configuration `cfg` (index into the planner's table) and operand layout
template <int EPI>
void launch_layout(hipStream_t st, const GemmParams &p, int cfg, bool vec, bool a_kmajor, bool b_kmajor);
// sums split-K slabs and applies epilogue EPI
template <int EPI>
void launch_reduce(hipStream_t st, const GemmParams &p, const float *slabs, int ld_slab);

#ifdef KHG_DEFINE_KERNELS
// Launches of the product kernels.  While a launch is being timed (hipK_gemm_timing), a start and a stop event are
// BOUND to the dispatch (hipExtLaunchKernelGGL): they read the kernel's own begin and end stamps -- what a kernel trace
// reports -- and put no packet of their own into the stream's queue.
extern hipEvent_t g_bind_start, g_bind_stop;
#define KHG_LAUNCH(kern, grid, block, lds, st, ...)                                                                 \
  do {                                                                                                              \
    if (khg::g_bind_stop != nullptr)                                                                                \
      hipExtLaunchKernelGGL(kern, grid, block, lds, st, khg::g_bind_start, khg::g_bind_stop, 0, __VA_ARGS__);       \
    else                                                                                                            \
      hipLaunchKernelGGL(kern, grid, block, lds, st, __VA_ARGS__);                                                  \
  } while (0)
#include <hip/hip_ext.h>

#ifdef KH_GEMM_STAMPS
// diagnostic build only: cycle stamps of wave 0 of every workgroup, 16 slots per workgroup
#define KH_STAMP(i)                                                                     \
  do {                                                                                  \
    if (p.stamps && threadIdx.x == 0) {                                                 \
      p.stamps[(size_t)blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memtime();           \
      /* slots 8 / 9: the 100 MHz real-time counter (common to all XCDs) at entry and exit */ \
      if ((i) == 0) p.stamps[(size_t)blockIdx.x * 16 + 8] = __builtin_amdgcn_s_memrealtime(); \
      if ((i) == 7) p.stamps[(size_t)blockIdx.x * 16 + 9] = __builtin_amdgcn_s_memrealtime(); \
    }                                                                                   \
  } while (0)
// real-time stamp into slot i (10..15) of the workgroup's 16
#define KH_RT(i)                                                                                          \
  do {                                                                                                    \
    if (p.stamps && threadIdx.x == 0) p.stamps[(size_t)blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
// cycle stamp into slot i (14, 15) of the workgroup's 16
#define KH_CYC(i)                                                                                     \
  do {                                                                                                \
    if (p.stamps && threadIdx.x == 0) p.stamps[(size_t)blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define KH_STAMP(i) do { } while (0)
#define KH_RT(i) do { } while (0)
#define KH_CYC(i) do { } while (0)
#endif

// The epilogue's second operand of one output element, fetched before any store so that the
// loads of all elements of a lane are in flight together: the bias of the column, or y.  Row and
// column are clamped into the matrix (elements outside are computed but never stored).
template <int EPI>
__device__ __forceinline__ float epi_aux(const GemmParams &p, int row, int col) {
  if (EPI == EPI_BIAS_ACT || EPI == EPI_BIAS_SAMPLE) return p.bias[min(col, p.N - 1)];
  if (EPI == EPI_DSIG) return p.y[(size_t)min(row, p.M - 1) * p.ldy + min(col, p.N - 1)];
  return 0.f;
}

// The arithmetic of the epilogue that needs no memory: done for every element of a lane before
// the guarded stores, so the wait for the operand loads happens once, not in front of every store
// (where it would also wait for the store before it).
template <int EPI>
__device__ __forceinline__ float epi_value(const GemmParams &p, float acc, float aux) {
  if (EPI == EPI_BIAS_ACT) {
    const float v = acc + aux;
    return p.act == KH_ACT_SIGMOID ? kh::sigmoidf(v) : v;
  }
  if (EPI == EPI_DSIG) return aux * (1.0f - aux) * acc;
  if (EPI == EPI_BIAS_SAMPLE) return kh::sigmoidf(acc + aux);
  return acc;
}

template <int EPI, bool SC1 = false>
__device__ __forceinline__ void epi_store(const GemmParams &p, int row, int col, float val, int ks) {
  if (EPI == EPI_SLAB) {
    p.C[(size_t)ks * p.slab_stride + (size_t)row * p.ldc + col] = val;
  } else if (EPI == EPI_AXPBY) {
    float *c = p.C + (size_t)row * p.ldc + col;
    *c = (p.beta == 0.f) ? p.alpha * val : p.alpha * val + p.beta * *c;
  } else if (EPI == EPI_BIAS_ACT || EPI == EPI_DSIG) {
#if defined(KH_CHAIN_PROBE) && KH_CHAIN_PROBE >= 2
    __hip_atomic_store(p.C + (size_t)row * p.ldc + col, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (sc1 store)
#else
    if (SC1) __hip_atomic_store(p.C + (size_t)row * p.ldc + col, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else p.C[(size_t)row * p.ldc + col] = val;
#endif
  } else if (EPI == EPI_UPDATE) {
    float c = p.alpha * val;
    if (p.mom != 0.f) c = kh::mul_add(p.mom, p.C[(size_t)row * p.ldc + col], c);
    if (p.C) p.C[(size_t)row * p.ldc + col] = c;
    float wv = p.W[(size_t)row * p.ldw + col];
    if (p.l2 != 0.f) wv = -p.l2 * wv + wv;
    p.Wout[(size_t)row * p.ldwout + col] = -p.lr * c + wv;
  } else if (EPI == EPI_RBM) {
    float *dw = p.C + (size_t)row * p.ldc + col;
    const float wv = p.W[(size_t)row * p.ldw + col];
    float d = p.alpha * val + *dw;
    d = -p.l2 * wv + d;
    *dw = d;
    p.Wout[(size_t)row * p.ldwout + col] = d + wv;
  } else if (EPI == EPI_GRAD) {
    p.C[(size_t)row * p.ldc + col] = p.alpha * val;
  } else if (EPI == EPI_BIAS_SAMPLE) {
    // (the form with one round trip per element; the tiles whose waves split K fetch the states of all their
    // elements together, see the epilogue)
    const size_t idx = (size_t)row * p.ldc + col;
    unsigned a = p.z1[idx], b = p.z2[idx], c = p.z3[idx], d = p.z4[idx];
    p.C[idx] = val;
    p.C2[idx] = (val > kh::hybrid_taus(a, b, c, d)) ? 1.0f : 0.0f;
    p.z1[idx] = a; p.z2[idx] = b; p.z3[idx] = c; p.z4[idx] = d;
  }
}

// Both statistics of the RBM step on one element (GemmParams::nk1): the arithmetic of an EPI_AXPBY launch with the
// negative statistics followed by an EPI_RBM launch with the positive ones, element for element.
__device__ __forceinline__ void rbm_dual_store(const GemmParams &p, int row, int col, float v1, float v2) {
  float *dw = p.C + (size_t)row * p.ldc + col;
  const float wv = p.W[(size_t)row * p.ldw + col];
  float d = (p.beta == 0.f) ? p.alpha * v1 : p.alpha * v1 + p.beta * *dw;
  d = p.alpha2 * v2 + d;
  d = -p.l2 * wv + d;
  *dw = d;
  p.Wout[(size_t)row * p.ldwout + col] = d + wv;
}

// BM x BN output tile, 4 waves arranged WAVES_M x WAVES_N x WK.  A K-tile is BKT = BK*WK deep and
// wave-slice wk of it (BK = 32 k) goes to the waves with that wk: with WK > 1 the waves of a
// workgroup split K among themselves and add their accumulators through LDS before the epilogue.
// That keeps 256 workgroups (one per CU) on a 256-row x 2048-column product with 32x64 tiles and
// the whole K inside the workgroup, so the epilogue can finish the element (bias, sigmoid, ...)
// instead of writing a partial slab for a second kernel.
// A_KMAJOR: A(m,k) at A[m*lda+k] (k contiguous), otherwise A[k*lda+m].  Same for B with n.
// VEC: every row base is 16-byte aligned and leading dimensions are multiples of 4 floats.
// floats of LDS one workgroup of gemm_tile<...> needs (two buffers of an A image and a B image)
template <int BM, int BN, int BK, int WK, bool A_KMAJOR, bool B_KMAJOR>
constexpr int gemm_lds_floats() {
  constexpr int BKT = BK * WK;
  constexpr int PA = A_KMAJOR ? (BKT + 4) : BM, PB = B_KMAJOR ? (BKT + 4) : BN;
  return 2 * ((A_KMAJOR ? BM * PA : BKT * PA) + (B_KMAJOR ? BN * PB : BKT * PB));
}

// One output tile by one workgroup: tile `bid` of the `nblk` this product is cut into (`bid` must be
// congruent to blockIdx.x modulo 8 for the XCD-aware decode below to keep its meaning).
// NW = 8 (only with WK = 1): a 512-thread workgroup, waves arranged 2 x 4 over the tile.
// KTAIL: the variant for K ranges that end inside a tile (see mask_tail below).
// CHAIN: this product runs inside a launch that also holds its producer (gemm_chain_kernel): operand A is loaded past
// the L1 (sc1) and the results are stored write-through (sc1), the two halves of a hand-off between compute units.
template <int BM, int BN, int BK, int WK, bool A_KMAJOR, bool B_KMAJOR, int EPI, bool VEC, int NW = 4, bool KTAIL = false, bool CHAIN = false>
__device__ __forceinline__ void gemm_tile(const GemmParams &p, float *lds, const int bid, const int nblk) {
  constexpr int NT = 64 * NW;
  static_assert(NW == 4 || (NW == 8 && WK == 1), "4 waves, or 8 waves without a K split over waves");
  constexpr int WAVES_M = (WK == 1) ? 2 : 1, WAVES_N = (NW == 8) ? 4 : (WK == 4) ? 1 : 2;
  static_assert(WAVES_M * WAVES_N * WK == NW, "waves per workgroup");
  constexpr int BKT = BK * WK;                          // K depth of one staged tile
  constexpr int TM = BM / (WAVES_M * 32), TN = BN / (WAVES_N * 32);
  static_assert(TM >= 1 && TN >= 1, "tile too small for this wave arrangement");
  constexpr int PA = A_KMAJOR ? (BKT + 4) : BM;        // LDS pitch of the A image (floats)
  constexpr int PB = B_KMAJOR ? (BKT + 4) : BN;
  constexpr int A_ELEMS = A_KMAJOR ? BM * PA : BKT * PA;
  constexpr int B_ELEMS = B_KMAJOR ? BN * PB : BKT * PB;
  constexpr int A_UNITS = BM * BKT / 4 / NT;           // float4 loads per thread per K-tile
  constexpr int B_UNITS = BN * BKT / 4 / NT;
  static_assert(A_UNITS >= 1 && B_UNITS >= 1, "tile too small for 256 threads");
  static_assert(2 * (A_ELEMS + B_ELEMS) == gemm_lds_floats<BM, BN, BK, WK, A_KMAJOR, B_KMAJOR>(), "LDS size helper out of step");

  // ---- which tile: XCD-aware decode of the linear block id -----------------------------------
  // Blocks b and b+8 run on the same XCD (same L2); give each XCD a contiguous run of tiles so
  // that neighbours, which share the B (weight) panel, hit in that L2.
  KH_STAMP(0);
  // The products of the dependent chain (forward, back-propagation) outrank the weight updates that
  // fill the compute units beside them: where waves of both share a SIMD, the chain's issue first.
#ifndef KH_NO_CHAIN_PRIO
  if (EPI != EPI_UPDATE && EPI != EPI_RBM && EPI != EPI_GRAD) __builtin_amdgcn_s_setprio(3);  // (EPI_BIAS_SAMPLE included)
#endif
  // Which way the run walks decides which operand an XCD reads whole: down the rows first it covers every row
  // tile of a few column tiles (all of A, a slice of B); along the columns first the other way round -- the host
  // picks the order that leaves the larger operand sliced (n_fastest; only without a K split).
  int L = bid;
  if ((nblk & 7) == 0) L = (L & 7) * (nblk >> 3) + (L >> 3);
  int mi, ni, ks;
  if (p.n_fastest) {
    ni = L % p.tiles_n;
    mi = L / p.tiles_n;
    ks = 0;
  } else {
    mi = L % p.tiles_m;
    L /= p.tiles_m;
    ks = L % p.splitk;
    ni = L / p.splitk;
  }
  const int m0 = mi * BM, n0 = ni * BN;
  const int kbeg = ks * p.k_per_split;
  const int kend = min(p.K, kbeg + p.k_per_split);
  const int nkt = (kend - kbeg + BKT - 1) / BKT;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wk = wid / (WAVES_M * WAVES_N);  // which K slice of every tile this wave multiplies
  const int wm0 = ((wid / WAVES_N) % WAVES_M) * (TM * 32), wn0 = (wid % WAVES_N) * (TN * 32);

  float4 ra[A_UNITS], rb[B_UNITS];

  // Loads of one K-tile into registers, two forms.
  // fast: VEC layout and every k of the tile inside [kbeg,kend).  Unconditional 16-byte loads from
  //   per-thread base pointers -- rows / columns past the matrix edge are redirected to the last
  //   valid one, their products only reach output elements that are never stored -- so all loads
  //   of a tile are in flight together and no wait sits between them.
  // slow: guarded element-wise loads with zero fill (the K-tail tile, unaligned operands).
  // The fast loads are buffer loads: one resource descriptor per operand in SGPRs, a 32-bit byte
  // offset per thread and unit, the K-tile advance as a scalar offset.  Measured beside MFMAs,
  // global_load with 64-bit per-lane addresses roughly doubles the cost of a K-tile; the buffer
  // form is nearly free (tools/mfma_probe2.hip), and out-of-range offsets return 0 instead of faulting.
  typedef int v4i __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void *)p.A, 0, p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc((void *)p.B, 0, p.b_bytes, 0x00020000);
  int oa[A_UNITS], ob[B_UNITS];  // byte offsets of this thread's units in K-tile 0
  int a_step, b_step;            // byte advance per K-tile
  {
    const int a_xlast = ((p.M - 1) >> 2) << 2, b_xlast = ((p.N - 1) >> 2) << 2;
#pragma unroll
    for (int u = 0; u < A_UNITS; ++u) {
      const int unit = tid + u * NT;
      if (A_KMAJOR) {
        const int row = unit / (BKT / 4), kq = unit % (BKT / 4);
        oa[u] = (min(m0 + row, p.M - 1) * p.lda + (kbeg + kq * 4)) * 4;
      } else {
        const int krow = unit / (BM / 4), mq = unit % (BM / 4);
        oa[u] = ((kbeg + krow) * p.lda + min(m0 + mq * 4, a_xlast)) * 4;
      }
    }
#pragma unroll
    for (int u = 0; u < B_UNITS; ++u) {
      const int unit = tid + u * NT;
      if (B_KMAJOR) {
        const int row = unit / (BKT / 4), kq = unit % (BKT / 4);
        ob[u] = (min(n0 + row, p.N - 1) * p.ldb + (kbeg + kq * 4)) * 4;
      } else {
        const int krow = unit / (BN / 4), nq = unit % (BN / 4);
        ob[u] = ((kbeg + krow) * p.ldb + min(n0 + nq * 4, b_xlast)) * 4;
      }
    }
    a_step = (A_KMAJOR ? BKT : BKT * p.lda) * 4;
    b_step = (B_KMAJOR ? BKT : BKT * p.ldb) * 4;
  }
  auto buf_load = [&](const __amdgpu_buffer_rsrc_t &rs, int voff, int soff) {
    const v4i r = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
    return make_float4(__int_as_float(r.x), __int_as_float(r.y), __int_as_float(r.z), __int_as_float(r.w));
  };
#ifdef KH_CHAIN_PROBE
  // Diagnostic build (make alt ALT=-DKH_CHAIN_PROBE=1 or =2): what handing the activations from one layer's workgroups to the
  // next layer's INSIDE one launch would cost the kernels themselves -- the activation operand (A of the forward and
  // back-propagation products) is loaded past the L1 (sc1) and, with =2, the outputs are stored write-through (sc1; 4 bytes a
  // lane in this epilogue's layout), as such a hand-off needs; the weight panels stay on the cached path.  No synchronisation is added: the launches stay.
  auto buf_load_a = [&](const __amdgpu_buffer_rsrc_t &rs, int voff, int soff) {
    if (EPI != EPI_BIAS_ACT && EPI != EPI_DSIG) return buf_load(rs, voff, soff);
    const v4i r = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 16);
    return make_float4(__int_as_float(r.x), __int_as_float(r.y), __int_as_float(r.z), __int_as_float(r.w));
  };
#else
  auto buf_load_a = [&](const __amdgpu_buffer_rsrc_t &rs, int voff, int soff) {
    const v4i r = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, CHAIN ? 16 : 0);
    return make_float4(__int_as_float(r.x), __int_as_float(r.y), __int_as_float(r.z), __int_as_float(r.w));
  };
#endif
  // two products in one pass (EPI_RBM, p.nk1 > 0): K-tile kt >= nk1 is tile kt - nk1 of the second pair of operands
  constexpr bool DUAL_EPI = (EPI == EPI_RBM);
  const int nk1 = DUAL_EPI ? p.nk1 : 0;
  const __amdgpu_buffer_rsrc_t rsrc_a2 = __builtin_amdgcn_make_buffer_rsrc((void *)(DUAL_EPI ? p.A2 : p.A), 0, p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_b2 = __builtin_amdgcn_make_buffer_rsrc((void *)(DUAL_EPI ? p.B2 : p.B), 0, p.b_bytes, 0x00020000);
  // KTAIL: the K tail (a last tile that [kbeg, kend) fills only in part) as one more tile of the fast loop.  Before
  // the tail tile's loads are issued, the offsets of the units whose k lies past the end are replaced by one beyond
  // the operand's footprint, which a buffer load answers with 0, so the tile's remaining products add 0 -- the same
  // sums as the guarded tail.  The replacement runs once, under a uniform branch.  Needs whole 16-byte units on
  // either side of the end where K is the contiguous direction (K a multiple of 4; rows of k are whole units).
  // A variant of its own, launched only for such K: in the common kernel the mere presence of this code cost the
  // launches without a tail 0.4-0.5 us each (another register allocation), and selecting the offset beside every
  // load 7 % of a whole training step.  The guarded element-wise tail -- loads, then an unpipelined tile -- takes
  // 5.8 k cycles where a full tile takes 1.0 k (32 x 64 tile, K split over two waves).
  static_assert(!KTAIL || (NT % (BKT / 4) == 0 && NT % (BM / 4) == 0 && NT % (BN / 4) == 0), "unit -> k mapping below");
  bool tail_fast = false;
  if constexpr (KTAIL)
    tail_fast = VEC && p.a_bytes > 0 && ((kend - kbeg) % BKT) != 0 && (((A_KMAJOR || B_KMAJOR) ? ((kend - kbeg) & 3) : 0) == 0);
  auto mask_tail = [&]() {
    if constexpr (KTAIL) {
      if (!tail_fast) return;
      // k (inside a tile) of this thread's unit 0 of either operand; unit u lies a compile-time step further
      const int ka0 = A_KMAJOR ? (tid % (BKT / 4)) * 4 : tid / (BM / 4);
      const int kb0 = B_KMAJOR ? (tid % (BKT / 4)) * 4 : tid / (BN / 4);
      constexpr int KA_STEP = A_KMAJOR ? 0 : NT / (BM / 4), KB_STEP = B_KMAJOR ? 0 : NT / (BN / 4);
      const int rem = (kend - kbeg) % BKT;
#pragma unroll
      for (int u = 0; u < A_UNITS; ++u)
        if (ka0 + u * KA_STEP >= rem) oa[u] = p.a_bytes;
#pragma unroll
      for (int u = 0; u < B_UNITS; ++u)
        if (kb0 + u * KB_STEP >= rem) ob[u] = p.b_bytes;
    }
  };
  auto load_fast = [&](int kt) {
    if (DUAL_EPI && nk1 > 0 && kt >= nk1) {
#pragma unroll
      for (int u = 0; u < A_UNITS; ++u) ra[u] = buf_load(rsrc_a2, oa[u], (kt - nk1) * a_step);
#pragma unroll
      for (int u = 0; u < B_UNITS; ++u) rb[u] = buf_load(rsrc_b2, ob[u], (kt - nk1) * b_step);
      return;
    }
#pragma unroll
    for (int u = 0; u < A_UNITS; ++u) ra[u] = buf_load_a(rsrc_a, oa[u], kt * a_step);
#pragma unroll
    for (int u = 0; u < B_UNITS; ++u) rb[u] = buf_load(rsrc_b, ob[u], kt * b_step);
  };
  auto load_slow = [&](int kt) {
    const int k0 = kbeg + kt * BKT;
#pragma unroll
    for (int u = 0; u < A_UNITS; ++u) {
      const int unit = tid + u * NT;
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      if (A_KMAJOR) {
        const int row = unit / (BKT / 4), kq = unit % (BKT / 4);
        const int m = m0 + row, k = k0 + kq * 4;
        if (m < p.M) {
          const float *src = p.A + (size_t)m * p.lda + k;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (k + e < kend) v[e] = src[e];
        }
      } else {
        const int krow = unit / (BM / 4), mq = unit % (BM / 4);
        const int k = k0 + krow, m = m0 + mq * 4;
        if (k < kend) {
          const float *src = p.A + (size_t)k * p.lda + m;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (m + e < p.M) v[e] = src[e];
        }
      }
      ra[u] = make_float4(v[0], v[1], v[2], v[3]);
    }
#pragma unroll
    for (int u = 0; u < B_UNITS; ++u) {
      const int unit = tid + u * NT;
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      if (B_KMAJOR) {
        const int row = unit / (BKT / 4), kq = unit % (BKT / 4);
        const int n = n0 + row, k = k0 + kq * 4;
        if (n < p.N) {
          const float *src = p.B + (size_t)n * p.ldb + k;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (k + e < kend) v[e] = src[e];
        }
      } else {
        const int krow = unit / (BN / 4), nq = unit % (BN / 4);
        const int k = k0 + krow, n = n0 + nq * 4;
        if (k < kend) {
          const float *src = p.B + (size_t)k * p.ldb + n;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (n + e < p.N) v[e] = src[e];
        }
      }
      rb[u] = make_float4(v[0], v[1], v[2], v[3]);
    }
  };
  auto store_tiles = [&](int buf) {
    float *As = lds + buf * (A_ELEMS + B_ELEMS);
    float *Bs = As + A_ELEMS;
#pragma unroll
    for (int u = 0; u < A_UNITS; ++u) {
      const int unit = tid + u * NT;
      if (A_KMAJOR) {
        const int row = unit / (BKT / 4), kq = unit % (BKT / 4);
        *reinterpret_cast<float4 *>(As + row * PA + kq * 4) = ra[u];
      } else {
        const int krow = unit / (BM / 4), mq = unit % (BM / 4);
        *reinterpret_cast<float4 *>(As + krow * PA + mq * 4) = ra[u];
      }
    }
#pragma unroll
    for (int u = 0; u < B_UNITS; ++u) {
      const int unit = tid + u * NT;
      if (B_KMAJOR) {
        const int row = unit / (BKT / 4), kq = unit % (BKT / 4);
        *reinterpret_cast<float4 *>(Bs + row * PB + kq * 4) = rb[u];
      } else {
        const int krow = unit / (BN / 4), nq = unit % (BN / 4);
        *reinterpret_cast<float4 *>(Bs + krow * PB + nq * 4) = rb[u];
      }
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // first product's sums once the K loop has moved on to the second pair of operands (two products in one pass)
  f32x16 acc1[DUAL_EPI ? TM : 1][DUAL_EPI ? TN : 1];
  auto next_product = [&](int kt_now) {
    if (DUAL_EPI && nk1 > 0 && kt_now == nk1) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc1[DUAL_EPI ? i : 0][DUAL_EPI ? j : 0] = acc[i][j];
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        }
    }
  };

  // ---- operand fragments ------------------------------------------------------------------------
  // One group = 8 k = 4 MFMA steps per accumulator.  A fragment is fetched by RN single LDS read
  // instructions (one 16-byte read per 32-row block of a K-contiguous operand, four 4-byte reads
  // otherwise), numbered so that the pipelined loop can place them one by one.
  struct Frag {
    float a[TM][4], b[TN][4];
  };
  constexpr int RN_A = A_KMAJOR ? TM : 4 * TM, RN_B = B_KMAJOR ? TN : 4 * TN, RN = RN_A + RN_B;
  const int a_lds = A_KMAJOR ? (wm0 + li) * PA + wk * BK + lh * 4 : (wk * BK + lh * 4) * PA + wm0 + li;
  const int b_lds = A_ELEMS + (B_KMAJOR ? (wn0 + li) * PB + wk * BK + lh * 4 : (wk * BK + lh * 4) * PB + wn0 + li);
  auto read_op = [&](Frag &f, const float *base, int g, int r) {
    if (r < RN_A) {
      if (A_KMAJOR) {
        const float4 t = *reinterpret_cast<const float4 *>(base + a_lds + r * 32 * PA + g * 8);
        f.a[r][0] = t.x; f.a[r][1] = t.y; f.a[r][2] = t.z; f.a[r][3] = t.w;
      } else {
        f.a[r >> 2][r & 3] = base[a_lds + (g * 8 + (r & 3)) * PA + (r >> 2) * 32];
      }
    } else {
      r -= RN_A;
      if (B_KMAJOR) {
        const float4 t = *reinterpret_cast<const float4 *>(base + b_lds + r * 32 * PB + g * 8);
        f.b[r][0] = t.x; f.b[r][1] = t.y; f.b[r][2] = t.z; f.b[r][3] = t.w;
      } else {
        f.b[r >> 2][r & 3] = base[b_lds + (g * 8 + (r & 3)) * PB + (r >> 2) * 32];
      }
    }
  };
  // one register-staged 16-byte unit of the next tile: registers -> LDS, global -> registers
  constexpr int NC = A_UNITS + B_UNITS;
  auto write_op = [&](float *base, int c) {
    const float4 &va = ra[c < A_UNITS ? c : 0];
    const float4 &vb = rb[c < A_UNITS ? 0 : c - A_UNITS];
    if (c < A_UNITS) {
      const int unit = tid + c * NT;
      const int off = A_KMAJOR ? (unit / (BKT / 4)) * PA + (unit % (BKT / 4)) * 4 : (unit / (BM / 4)) * PA + (unit % (BM / 4)) * 4;
      *reinterpret_cast<float4 *>(base + off) = va;
    } else {
      const int unit = tid + (c - A_UNITS) * NT;
      const int off = B_KMAJOR ? (unit / (BKT / 4)) * PB + (unit % (BKT / 4)) * 4 : (unit / (BN / 4)) * PB + (unit % (BN / 4)) * 4;
      *reinterpret_cast<float4 *>(base + A_ELEMS + off) = vb;
    }
  };
  auto load_op = [&](int c, int kt) {
    float4 v;
    if (DUAL_EPI && nk1 > 0 && kt >= nk1)
      v = (c < A_UNITS) ? buf_load(rsrc_a2, oa[c < A_UNITS ? c : 0], (kt - nk1) * a_step)
                        : buf_load(rsrc_b2, ob[c < A_UNITS ? 0 : c - A_UNITS], (kt - nk1) * b_step);
    else
      v = (c < A_UNITS) ? buf_load_a(rsrc_a, oa[c < A_UNITS ? c : 0], kt * a_step)
                        : buf_load(rsrc_b, ob[c < A_UNITS ? 0 : c - A_UNITS], kt * b_step);
    if (c < A_UNITS) ra[c] = v; else rb[c - A_UNITS] = v;
  };
  // MFMAs of one K-tile out of LDS buffer `buf`, unpipelined (K tail, unaligned operands)
  auto compute = [&](int buf) {
    const float *base = lds + buf * (A_ELEMS + B_ELEMS);
#pragma unroll
    for (int g = 0; g < BK / 8; ++g) {
      Frag f;
#pragma unroll
      for (int r = 0; r < RN; ++r) read_op(f, base, g, r);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int q = 0; q < TN; ++q)
            acc[i][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[i][j], f.b[q][j], acc[i][q], 0, 0, 0);
    }
  };
  // Workgroup barrier that drains this wave's LDS traffic only: global loads issued for a later
  // K-tile stay in flight across it (__syncthreads would wait for them too).
  auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

  // Epilogue operand fetched under the main loop (the weight tile of the update GEMM): one
  // accumulator's worth per K-tile, so that the read-modify-write at the end starts with the old
  // weights already in registers instead of every workgroup hitting HBM at the same moment.
  constexpr bool PREFETCH = (EPI == EPI_UPDATE || EPI == EPI_RBM);  // (EPI_RBM: only in its 16-byte form, two products in one pass)
  static_assert(!(PREFETCH && WK > 1), "the update epilogue is not built for K split over waves");
  constexpr int NPRE = PREFETCH ? TM * TN : 1;
  float pre[NPRE][16];
  const bool full_tile = (m0 + BM <= p.M) && (n0 + BN <= p.N);
  int npre = 0;  // accumulators whose operand is in `pre`
  // Two layouts of one accumulator's worth of old weights in `pre`.  Plain: register r = the element of accumulator
  // register r (16 four-byte loads).  Vector (p.w_bytes != 0): registers 4*pass .. 4*pass+3 = four consecutive
  // columns of row (lane / 8 + 8 * pass) of the 32 x 32 block, one 16-byte buffer load each; rows past the matrix
  // read 0 (buffer bounds), so edge tiles take this form too.
  const bool epi_vec = PREFETCH && p.w_bytes != 0;
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void *)p.W, 0, p.w_bytes, 0x00020000);
  auto vec_off = [&](int a, int pass, int ld) {  // byte offset of this lane's 4 columns of row `pass` of accumulator a
    const int i = a / TN, q = a % TN;
    return ((m0 + wm0 + i * 32 + (lane >> 3) + 8 * pass) * ld + (n0 + wn0 + q * 32 + (lane & 7) * 4)) * 4;
  };
  auto prefetch_acc = [&](int a) {
    if (epi_vec) {
#pragma unroll
      for (int pass = 0; pass < 4; ++pass) {
        const float4 t = buf_load(rsrc_w, vec_off(a, pass, p.ldw), 0);
        pre[a][4 * pass] = t.x; pre[a][4 * pass + 1] = t.y; pre[a][4 * pass + 2] = t.z; pre[a][4 * pass + 3] = t.w;
      }
      return;
    }
    const int i = a / TN, q = a % TN;
    const float *src = p.W + (size_t)(m0 + wm0 + i * 32 + 4 * lh) * p.ldw + (n0 + wn0 + q * 32 + li);
#pragma unroll
    for (int r = 0; r < 16; ++r) pre[a][r] = src[(size_t)((r & 3) + 8 * (r >> 2)) * p.ldw];
  };

  // K-tiles completely inside the split take the fast loads; at most one tail tile does not.
  const int nfull = (VEC && p.a_bytes > 0) ? ((KTAIL && tail_fast) ? nkt : (kend - kbeg) / BKT) : 0;
  constexpr int G = BK / 8;           // k-groups per tile
  // which K-loop: whole-tile operand fragments (more registers: 2 waves per SIMD at most) or
  // group-by-group fragments (the update GEMMs, which run 4 workgroups per CU)
  constexpr bool kWholeTile = (KH_SCHED == 2) || (KH_SCHED == 3 && EPI != EPI_UPDATE && EPI != EPI_RBM && EPI != EPI_GRAD);
  constexpr int MG = 4 * TM * TN;     // MFMAs per group
  static_assert(G >= 4 && G % 2 == 0, "BK must be a multiple of 16, at least 32");
  int kt = 0;
  if (nfull == 0) {
    // unaligned operands or a K range shorter than one tile: plain double-buffered loop
    if (nkt > 0) {
      load_slow(0);
      store_tiles(0);
    }
    __syncthreads();
    for (; kt < nkt; ++kt) {
      const bool more = (kt + 1 < nkt);
      if (more) load_slow(kt + 1);
      compute(kt & 1);
      if (more) store_tiles((kt + 1) & 1);
      __syncthreads();
    }
  } else {
    // Software pipeline inside the wave (one wave per SIMD has nobody else to hide its latencies):
    // the memory instructions of the NEXT tiles are placed one by one behind the MFMAs of this
    // one and pinned there with scheduling barriers, so that they issue in the MFMAs' shadow.
    // Measured beside a chain of MFMAs (tools/mfma_probe2.hip, mfma_probe4.hip): a gap between two
    // MFMAs that holds other instructions costs ~8 cycles whatever is in it (as long as it issues
    // within the MFMA's 64 cycles), an empty gap nothing.  Two forms follow.
    if constexpr (kWholeTile) {
    // Whole-tile operand fragments.  The operands of ALL k-groups of a tile are held in registers,
    // double-buffered across tiles, so a tile is NM = G*MG MFMAs in a row with only
    //   * one gap per copy unit (LDS store of the next tile + buffer load for the tile after), and
    //   * behind MFMA SB: the wait for this wave's LDS stores, the barrier, and the LDS reads of
    //     the whole next tile (8 read instructions per gap, consecutive gaps).
    // Measured (tools/mfma_probe4.hip): 1068 cycles per 64x64x32 tile against 1024 of MFMA, where
    // reading group by group takes ~1250.
    if constexpr (KTAIL) {
      if (nfull == 1) mask_tail();
    }
    load_fast(0);
    KH_STAMP(1);
    store_tiles(0);
    KH_STAMP(2);
    __syncthreads();
    KH_STAMP(3);
    if constexpr (KTAIL) {
      if (nfull == 2) mask_tail();
    }
    if (nfull > 1) load_fast(1);
    Frag fr[2][G];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int r = 0; r < RN; ++r) read_op(fr[0][g], lds, g, r);
    constexpr int NM = G * MG;            // MFMAs per tile
    constexpr int NREAD = G * RN;         // LDS read instructions per tile
    constexpr int RGAPS = (NREAD + 7) / 8;
    constexpr int SB = NM - 2 - RGAPS;    // MFMA slot behind which the barrier and the first reads go
    static_assert(SB >= 1 + NC, "tile too small for this schedule");
    auto tile_body = [&](auto last_tag, auto par_tag) {
      constexpr bool LAST = decltype(last_tag)::value;
      constexpr int PAR = decltype(par_tag)::value;  // = kt & 1: LDS buffer and fragment set of this tile
      float *nxt = lds + (PAR ^ 1) * (A_ELEMS + B_ELEMS);
      const int kt_load = min(kt + 2, nfull - 1);  // past the end: a harmless repeat of the last tile
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        const int g = m / MG, t = m % MG;
        const int j = t / (TM * TN), i = (t % (TM * TN)) / TN, q = t % TN;
        acc[i][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(fr[PAR][g].a[i][j], fr[PAR][g].b[q][j], acc[i][q], 0, 0, 0);
        if (!LAST) {
#pragma unroll
          for (int c = 0; c < NC; ++c)
            if (1 + (c * (SB - 1)) / NC == m) {
#if defined(KH_GEMM_STAMPS) && defined(KH_ABL)
              // timing ablations of the diagnostic build (wrong results on purpose): 1 no LDS stores, 2 no loads, 3 neither
              if (KH_ABL != 1 && KH_ABL != 3) write_op(nxt, c);
              if (KH_ABL != 2 && KH_ABL != 3) load_op(c, kt_load);
#else
              write_op(nxt, c);
              load_op(c, kt_load);
#endif
            }
          if (m >= SB && m < SB + RGAPS) {
#if defined(KH_GEMM_STAMPS) && defined(KH_ABL) && KH_ABL >= 4
            if (m == SB && KH_ABL != 4) lds_barrier();  // 4: no barrier
#else
            if (m == SB) lds_barrier();
#endif
#pragma unroll
            for (int x = (m - SB) * 8; x < (m - SB + 1) * 8 && x < NREAD; ++x)
              read_op(fr[PAR ^ 1][x / RN], nxt, x / RN, x % RN);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    auto steady = [&](auto par_tag) {
      if (PREFETCH && (EPI == EPI_UPDATE ? (full_tile || epi_vec) : epi_vec)) {
#pragma unroll
        for (int a = 0; a < NPRE; ++a)
          if (kt == a) {
            prefetch_acc(a);
            npre = a + 1;
          }
      }
      if constexpr (KTAIL) {
        if (kt == nfull - 3) mask_tail();  // this tile's body issues the first loads of the last tile
      }
      tile_body(std::false_type(), par_tag);
    };
    // two tiles per trip, no exit in the middle: the fragment sets and accumulators then keep
    // their registers around the back edge (a mid-loop exit made the compiler copy all of them)
    while (kt + 2 < nfull) {
      steady(std::integral_constant<int, 0>());
      ++kt;
      steady(std::integral_constant<int, 1>());
      ++kt;
    }
    const bool tail = nkt > nfull;  // the K tail, at most one partial tile
    if (kt + 1 < nfull) {           // kt is even here
      steady(std::integral_constant<int, 0>());
      ++kt;
      KH_STAMP(4);
      if (tail) load_slow(nfull);
      tile_body(std::true_type(), std::integral_constant<int, 1>());
    } else {
      KH_STAMP(4);
      if (tail) load_slow(nfull);
      tile_body(std::true_type(), std::integral_constant<int, 0>());
    }
    if (tail) {
      store_tiles(nfull & 1);
      __syncthreads();
      compute(nfull & 1);
    }
    } else {
    load_fast(0);
    KH_STAMP(1);
    store_tiles(0);
    KH_STAMP(2);
    __syncthreads();
    KH_STAMP(3);
    if (nfull > 1) load_fast(1);
    Frag fr[2];
#pragma unroll
    for (int r = 0; r < RN; ++r) read_op(fr[0], lds, 0, r);

    // Group-by-group fragments (the update GEMMs): two fragment buffers; the LDS reads of group
    // g+1 go behind the first MFMAs of group g, the copy units behind the MFMAs of the second half
    // of groups 1..G-2 (an LDS store never sits in the queue in front of the next group's
    // operand reads), the one barrier per tile behind the first MFMA of the last group.
    auto tile_body = [&](auto last_tag) {
      constexpr bool LAST = decltype(last_tag)::value;
      float *cur = lds + (kt & 1) * (A_ELEMS + B_ELEMS);
      float *nxt = lds + ((kt + 1) & 1) * (A_ELEMS + B_ELEMS);
      const int kt_load = min(kt + 2, nfull - 1);  // past the end: a harmless repeat of the last tile
#pragma unroll
      for (int g = 0; g < G; ++g) {
#pragma unroll
        for (int t = 0; t < MG; ++t) {
          const int j = t / (TM * TN), i = (t % (TM * TN)) / TN, q = t % TN;
          acc[i][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(fr[g & 1].a[i][j], fr[g & 1].b[q][j], acc[i][q], 0, 0, 0);
          // reads of group g+1 behind the first MFMAs of group g, copies in the second half of groups 1..G-2
          constexpr int RPS = (RN + (MG / 2) - 1) / (MG / 2);
          if (g + 1 < G) {
#pragma unroll
            for (int r = 0; r < RN; ++r)
              if (r / RPS == t) read_op(fr[(g + 1) & 1], cur, g + 1, r);
          } else if (!LAST) {
            if (t == 0) lds_barrier();
#pragma unroll
            for (int r = 0; r < RN; ++r)
              if (r / RPS == t) read_op(fr[0], nxt, 0, r);
          }
          if (!LAST && g >= 1 && g <= G - 2) {
            constexpr int PER = (NC + (G - 2) - 1) / (G - 2);
            constexpr int STEP = (MG / 2) / PER > 0 ? (MG / 2) / PER : 1;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
              const int tt = MG / 2 + (c % PER) * STEP;
              if ((1 + c / PER) * MG + (tt < MG ? tt : MG - 1) == g * MG + t) {
                write_op(nxt, c);
                load_op(c, kt_load);
              }
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    };
    auto steady = [&]() {
      if (PREFETCH && (EPI == EPI_UPDATE ? (full_tile || epi_vec) : epi_vec)) {
#pragma unroll
        for (int a = 0; a < NPRE; ++a)
          if (kt == a) {
            prefetch_acc(a);
            npre = a + 1;
          }
      }
      next_product(kt);
      tile_body(std::false_type());
    };
    for (; kt + 1 < nfull; ++kt) steady();
    KH_STAMP(4);
    const bool tail = nkt > nfull;  // the K tail, at most one partial tile
    if (tail) load_slow(nfull);
    next_product(kt);
    tile_body(std::true_type());
    if (tail) {
      store_tiles(nfull & 1);
      __syncthreads();
      compute(nfull & 1);
    }
    }
  }

  KH_STAMP(5);
  // ---- epilogue: accumulator register r of lane (li, lh) is C[(r&3) + 8*(r>>2) + 4*lh][li] ------
  if constexpr (WK == 1) {
    if (EPI == EPI_RBM && epi_vec) {
      // Vector form of the RBM epilogue, two products in one pass (the host sets the footprints only then): as the
      // update epilogue below -- accumulators turned through the wave's own region of LDS, 16-byte accesses -- with
      // both sets of sums and the arithmetic of rbm_dual_store.  The first set is turned and kept in registers, then
      // the second goes through the same region (LDS traffic of one wave is in order).
      typedef int v4i_t __attribute__((ext_vector_type(4)));
      const __amdgpu_buffer_rsrc_t rsrc_wout = __builtin_amdgcn_make_buffer_rsrc((void *)p.Wout, 0, p.wout_bytes, 0x00020000);
      const __amdgpu_buffer_rsrc_t rsrc_c = __builtin_amdgcn_make_buffer_rsrc((void *)p.C, 0, p.c_bytes, 0x00020000);
      float *xp = lds + wid * (32 * 36);
      lds_barrier();  // every wave has read its last operand fragments
#pragma unroll
      for (int a = 0; a < TM * TN; ++a) {
        if (a >= npre) prefetch_acc(a);
        float4 old[4];
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) old[pass] = buf_load(rsrc_c, vec_off(a, pass, p.ldc), 0);
        // one wait for every load of this accumulator's epilogue, before its first store (see the update epilogue)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int pass = 0; pass < 4; ++pass)
          asm volatile("" : "+v"(old[pass].x), "+v"(old[pass].y), "+v"(old[pass].z), "+v"(old[pass].w));
#pragma unroll
        for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(pre[PREFETCH ? a : 0][r]));
        float4 v1[4];
#pragma unroll
        for (int r = 0; r < 16; ++r) xp[((r & 3) + 8 * (r >> 2) + 4 * lh) * 36 + li] = acc1[DUAL_EPI ? a / TN : 0][DUAL_EPI ? a % TN : 0][r];
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) v1[pass] = *reinterpret_cast<const float4 *>(xp + ((lane >> 3) + 8 * pass) * 36 + (lane & 7) * 4);
#pragma unroll
        for (int r = 0; r < 16; ++r) xp[((r & 3) + 8 * (r >> 2) + 4 * lh) * 36 + li] = acc[a / TN][a % TN][r];
        const int col = n0 + wn0 + (a % TN) * 32 + (lane & 7) * 4;
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
          const float4 v2 = *reinterpret_cast<const float4 *>(xp + ((lane >> 3) + 8 * pass) * 36 + (lane & 7) * 4);
          const float a1[4] = {v1[pass].x, v1[pass].y, v1[pass].z, v1[pass].w}, a2[4] = {v2.x, v2.y, v2.z, v2.w};
          const float o[4] = {old[pass].x, old[pass].y, old[pass].z, old[pass].w};
          float d[4], w[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float wv = pre[PREFETCH ? a : 0][4 * pass + e];
            float t = (p.beta == 0.f) ? p.alpha * a1[e] : p.alpha * a1[e] + p.beta * o[e];
            t = p.alpha2 * a2[e] + t;
            t = -p.l2 * wv + t;
            d[e] = t;
            w[e] = t + wv;
          }
          if (col < p.N) {  // (rows past the matrix are dropped by the buffer bounds)
            const v4i_t dv = {__float_as_int(d[0]), __float_as_int(d[1]), __float_as_int(d[2]), __float_as_int(d[3])};
            __builtin_amdgcn_raw_buffer_store_b128(dv, rsrc_c, vec_off(a, pass, p.ldc), 0, 0);
            const v4i_t wv4 = {__float_as_int(w[0]), __float_as_int(w[1]), __float_as_int(w[2]), __float_as_int(w[3])};
            __builtin_amdgcn_raw_buffer_store_b128(wv4, rsrc_wout, vec_off(a, pass, p.ldwout), 0, KH_UPDATE_STORE_AUX);
          }
        }
      }
    } else if (EPI == EPI_UPDATE && epi_vec) {
      // Vector form of the update epilogue.  Each wave turns its accumulators, one 32 x 32 block at a time, through
      // a region of LDS of its own (the operand tiles are done with) so that a lane holds four consecutive columns
      // of a row: the old weights came in, and the new ones (and the momentum buffer) go out, 16 bytes per lane,
      // a wave instruction covering 8 whole 128-byte lines.  The new weights are stored write-through (sc1): they
      // are not read again before the next forward pass, and 17 MB of dirty lines in L2 would otherwise be written
      // back at the kernel's end, where the next launch waits for them (3-4 us per update launch measured).
      // Same arithmetic, element for element, as epi_store<EPI_UPDATE>.
      static_assert(!PREFETCH || NW * 32 * 36 <= 2 * (A_ELEMS + B_ELEMS), "transposition areas larger than the tile buffers");
      typedef int v4i_t __attribute__((ext_vector_type(4)));
      const __amdgpu_buffer_rsrc_t rsrc_wout = __builtin_amdgcn_make_buffer_rsrc((void *)p.Wout, 0, p.wout_bytes, 0x00020000);
      const __amdgpu_buffer_rsrc_t rsrc_c = __builtin_amdgcn_make_buffer_rsrc((void *)p.C, 0, p.c_bytes, 0x00020000);
      float *xp = lds + wid * (32 * 36);
      lds_barrier();  // every wave has read its last operand fragments
      KH_CYC(14);
      // Every load of the epilogue is waited for HERE, before its first store.  The memory counter retires loads
      // and stores in order, and the compiler, which cannot know that the old weights arrived during the K loop,
      // otherwise places a counted wait in front of each pass's use of them -- a wait that, from the third pass
      // on, also waits for the write-through stores of the passes before it (in-kernel stamps: ~1400 cycles per
      // store instruction, 11 k cycles for the eight of a 64 x 32 wave tile).  After this wait the registers are
      // "touched" by empty asm statements, so the compiler sees them as plain values.
#pragma unroll
      for (int a = 0; a < TM * TN; ++a)
        if (a >= npre) prefetch_acc(a);  // (K shorter than the number of accumulators)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int a = 0; a < TM * TN; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(pre[PREFETCH ? a : 0][r]));
#pragma unroll
      for (int a = 0; a < TM * TN; ++a) {
#pragma unroll
        for (int r = 0; r < 16; ++r) xp[((r & 3) + 8 * (r >> 2) + 4 * lh) * 36 + li] = acc[a / TN][a % TN][r];
        if (a == 0) {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          KH_CYC(15);
        }
        const int col = n0 + wn0 + (a % TN) * 32 + (lane & 7) * 4;
        // momentum: the old corrections of all four passes in flight together, one wait (which also waits for
        // the stores of the accumulator before this one: once per accumulator instead of once per pass)
        float4 oldc[4];
        if (p.mom != 0.f) {
#pragma unroll
          for (int pass = 0; pass < 4; ++pass) oldc[pass] = buf_load(rsrc_c, vec_off(a, pass, p.ldc), 0);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
          for (int pass = 0; pass < 4; ++pass)
            asm volatile("" : "+v"(oldc[pass].x), "+v"(oldc[pass].y), "+v"(oldc[pass].z), "+v"(oldc[pass].w));
        }
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
          const float4 v = *reinterpret_cast<const float4 *>(xp + ((lane >> 3) + 8 * pass) * 36 + (lane & 7) * 4);
          float c[4] = {p.alpha * v.x, p.alpha * v.y, p.alpha * v.z, p.alpha * v.w};
          if (p.mom != 0.f) {
            const float4 o = oldc[pass];
            c[0] = kh::mul_add(p.mom, o.x, c[0]); c[1] = kh::mul_add(p.mom, o.y, c[1]);
            c[2] = kh::mul_add(p.mom, o.z, c[2]); c[3] = kh::mul_add(p.mom, o.w, c[3]);
          }
          float w[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float wv = pre[PREFETCH ? a : 0][4 * pass + e];
            if (p.l2 != 0.f) wv = -p.l2 * wv + wv;
            w[e] = -p.lr * c[e] + wv;
          }
          if (col < p.N) {  // (rows past the matrix are dropped by the buffer bounds)
            if (p.C) {
              const v4i_t cv = {__float_as_int(c[0]), __float_as_int(c[1]), __float_as_int(c[2]), __float_as_int(c[3])};
              __builtin_amdgcn_raw_buffer_store_b128(cv, rsrc_c, vec_off(a, pass, p.ldc), 0, KH_UPDATE_STORE_AUX);
            }
            const v4i_t wv4 = {__float_as_int(w[0]), __float_as_int(w[1]), __float_as_int(w[2]), __float_as_int(w[3])};
            __builtin_amdgcn_raw_buffer_store_b128(wv4, rsrc_wout, vec_off(a, pass, p.ldwout), 0, KH_UPDATE_STORE_AUX);
          }
        }
      }
    } else if (EPI == EPI_GRAD && p.c_bytes != 0) {
      // The update epilogue's way out without its weight step: every 32 x 32 accumulator block turned through the
      // wave's own region of LDS so that a lane holds four consecutive columns of a row, stored 16 bytes per lane,
      // write-through (the next reader is the exchange between the ranks, not this compute unit).
      static_assert(NW * 32 * 36 <= 2 * (A_ELEMS + B_ELEMS), "transposition areas larger than the tile buffers");
      typedef int v4i_t __attribute__((ext_vector_type(4)));
      const __amdgpu_buffer_rsrc_t rsrc_c = __builtin_amdgcn_make_buffer_rsrc((void *)p.C, 0, p.c_bytes, 0x00020000);
      float *xp = lds + wid * (32 * 36);
      lds_barrier();  // every wave has read its last operand fragments
#pragma unroll
      for (int a = 0; a < TM * TN; ++a) {
#pragma unroll
        for (int r = 0; r < 16; ++r) xp[((r & 3) + 8 * (r >> 2) + 4 * lh) * 36 + li] = acc[a / TN][a % TN][r];
        const int col = n0 + wn0 + (a % TN) * 32 + (lane & 7) * 4;
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
          const float4 v = *reinterpret_cast<const float4 *>(xp + ((lane >> 3) + 8 * pass) * 36 + (lane & 7) * 4);
          if (col < p.N) {  // (rows past the matrix are dropped by the buffer bounds)
            const v4i_t cv = {__float_as_int(p.alpha * v.x), __float_as_int(p.alpha * v.y), __float_as_int(p.alpha * v.z),
                              __float_as_int(p.alpha * v.w)};
            __builtin_amdgcn_raw_buffer_store_b128(cv, rsrc_c, vec_off(a, pass, p.ldc), 0, KH_UPDATE_STORE_AUX);
          }
        }
      }
    } else
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int q = 0; q < TN; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          const int col = n0 + wn0 + q * 32 + li;
          if (EPI == EPI_UPDATE && i * TN + q < npre) {
            // full tile, old weight already here (same arithmetic as epi_store<EPI_UPDATE>)
            float c = p.alpha * acc[i][q][r];
            if (p.mom != 0.f) c = kh::mul_add(p.mom, p.C[(size_t)row * p.ldc + col], c);
            if (p.C) p.C[(size_t)row * p.ldc + col] = c;
            float wv = pre[PREFETCH ? i * TN + q : 0][r];
            if (p.l2 != 0.f) wv = -p.l2 * wv + wv;
            p.Wout[(size_t)row * p.ldwout + col] = -p.lr * c + wv;
          } else if (row < p.M && col < p.N) {
            if (DUAL_EPI && nk1 > 0) rbm_dual_store(p, row, col, acc1[DUAL_EPI ? i : 0][DUAL_EPI ? q : 0][r], acc[i][q][r]);
            else epi_store<EPI, CHAIN>(p, row, col, epi_value<EPI>(p, acc[i][q][r], epi_aux<EPI>(p, row, col)), ks);
          }
        }
  } else {
    // K split over the waves: wave wk finishes registers [wk*RS, (wk+1)*RS) of every accumulator.
    // Every wave puts all its registers into LDS (the tile buffers are free now), sorted by owner;
    // the owner adds the WK partial sums in the order of wk, so the result does not depend on
    // timing.  Register numbers stay compile-time constants this way (only addresses depend on
    // wk); selecting registers by wk made the compiler emit a 16-way select per element.
    // Exchange image: [spatial wave][owner][source][accumulator][register][lane].
    constexpr int RS = 16 / WK;
    constexpr int NACC = TM * TN;
    constexpr int XCH = WK * WK * NACC * RS * 64;  // floats per spatial wave position
    static_assert(WAVES_M * WAVES_N * XCH <= 2 * (A_ELEMS + B_ELEMS), "exchange image larger than the tile buffers");
    float *xch = lds + (wid % (WAVES_M * WAVES_N)) * XCH;
    lds_barrier();  // every wave is done reading operand tiles
#pragma unroll
    for (int o = 0; o < WK; ++o)
#pragma unroll
      for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int rr = 0; rr < RS; ++rr)
          xch[(((o * WK + wk) * NACC + a) * RS + rr) * 64 + lane] = acc[a / TN][a % TN][o * RS + rr];
    float aux[NACC][RS];
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
      for (int rr = 0; rr < RS; ++rr) {
        const int r = wk * RS + rr;
        aux[a][rr] = epi_aux<EPI>(p, m0 + wm0 + (a / TN) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, n0 + wn0 + (a % TN) * 32 + li);
      }
    // EPI_BIAS_SAMPLE: the generator states of this lane's elements, all in flight together with the bias (elements
    // past the matrix edge read the last valid one and are never stored)
    unsigned zs[EPI == EPI_BIAS_SAMPLE ? NACC : 1][EPI == EPI_BIAS_SAMPLE ? RS : 1][4];
    if (EPI == EPI_BIAS_SAMPLE) {
#pragma unroll
      for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int rr = 0; rr < RS; ++rr) {
          const int r = wk * RS + rr;
          const int row = min(m0 + wm0 + (a / TN) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, p.M - 1);
          const int col = min(n0 + wn0 + (a % TN) * 32 + li, p.N - 1);
          const size_t idx = (size_t)row * p.ldc + col;
          zs[EPI == EPI_BIAS_SAMPLE ? a : 0][EPI == EPI_BIAS_SAMPLE ? rr : 0][0] = p.z1[idx];
          zs[EPI == EPI_BIAS_SAMPLE ? a : 0][EPI == EPI_BIAS_SAMPLE ? rr : 0][1] = p.z2[idx];
          zs[EPI == EPI_BIAS_SAMPLE ? a : 0][EPI == EPI_BIAS_SAMPLE ? rr : 0][2] = p.z3[idx];
          zs[EPI == EPI_BIAS_SAMPLE ? a : 0][EPI == EPI_BIAS_SAMPLE ? rr : 0][3] = p.z4[idx];
        }
    }
    lds_barrier();  // (not __syncthreads: the loads just issued stay in flight across it)
    float sum[NACC][RS];
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
      for (int rr = 0; rr < RS; ++rr)
#pragma unroll
        for (int src = 0; src < WK; ++src) {
          const float part = xch[(((wk * WK + src) * NACC + a) * RS + rr) * 64 + lane];
          sum[a][rr] = (src == 0) ? part : sum[a][rr] + part;
        }
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
      for (int rr = 0; rr < RS; ++rr) sum[a][rr] = epi_value<EPI>(p, sum[a][rr], aux[a][rr]);
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
      for (int rr = 0; rr < RS; ++rr) {
        const int r = wk * RS + rr;
        const int row = m0 + wm0 + (a / TN) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int col = n0 + wn0 + (a % TN) * 32 + li;
        if (EPI == EPI_BIAS_SAMPLE) {
          unsigned *z = zs[EPI == EPI_BIAS_SAMPLE ? a : 0][EPI == EPI_BIAS_SAMPLE ? rr : 0];
          const float u = kh::hybrid_taus(z[0], z[1], z[2], z[3]);
          if (row < p.M && col < p.N) {
            const size_t idx = (size_t)row * p.ldc + col;
            p.C[idx] = sum[a][rr];
            p.C2[idx] = (sum[a][rr] > u) ? 1.0f : 0.0f;
            p.z1[idx] = z[0]; p.z2[idx] = z[1]; p.z3[idx] = z[2]; p.z4[idx] = z[3];
          }
        } else if (row < p.M && col < p.N) {
          epi_store<EPI, CHAIN>(p, row, col, sum[a][rr], ks);
        }
      }
  }
  KH_STAMP(6);
#ifdef KH_GEMM_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  KH_STAMP(7);
#endif
}

// Bias half of BiasedLinearity::Update for 64 outputs, by one 256-thread workgroup of the weight-update GEMM's
// launch (no launch of its own): bcorr = alpha * colsum(e) + mom * bcorr ; bias -= lr * bcorr.
// The sums are taken exactly like bias_update_kernel's (16 interleaved row groups, combined in order 0..15),
// so both give the same bits.
template <bool GRAD = false>
__device__ __forceinline__ void bias_update_tile(const GemmParams &p, float *lds, int tile) {
  KH_RT(10);
  const int tid = threadIdx.x;
  if (tid >= 256) {
    // (a 512-thread workgroup: the second half only keeps the barrier company)
  } else if (p.a_bytes != 0 && (p.lda & 3) == 0 && (reinterpret_cast<size_t>(p.A) & 15) == 0) {
    // Thread (g = tid / 16, quad = tid % 16) sums rows g, g + 16, g + 32, ... of 4 adjacent columns: one 16-byte
    // buffer load per row, the 16 rows of a 256-frame bunch all in flight, added in row order.  Rows past K lie
    // beyond the descriptor's range and load as 0; columns past M are summed (from padding or the next row)
    // and never stored.  Kept short on purpose: the workgroup runs this once, cold in the instruction cache,
    // beside workgroups that keep memory busy -- a fully unrolled form with a load per element (1200
    // instructions) spent 14 us fetching its own code there (in-kernel stamps), this one is ~100 instructions.
    typedef float v4f __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)p.A, 0, p.a_bytes, 0x00020000);
    const int g = tid >> 4, quad = tid & 15;
    const int voff = (g * p.lda + tile * 64 + quad * 4) * 4;
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    for (int r0 = 0; r0 < p.K; r0 += 256) {
      v4f x[16];
#pragma unroll
      for (int u = 0; u < 16; ++u)
        x[u] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, (r0 + 16 * u) * p.lda * 4, 0));
#pragma unroll
      for (int u = 0; u < 16; ++u) acc += x[u];
    }
    KH_RT(11);
    *reinterpret_cast<v4f *>(&lds[g * 64 + quad * 4]) = acc;
  } else {
    // any layout: a load per element, thread (q = tid / 64, lane) takes row groups q, q + 4, q + 8, q + 12 of column lane
    const int lane = tid & 63, q = tid >> 6;
    const int c = tile * 64 + lane;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (c < p.M) {
      const float *e = p.A + c;
      for (int r0 = q; r0 < p.K; r0 += 16 * 8) {
        float x[4][8];
#pragma unroll
        for (int v = 0; v < 4; ++v)
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int r = r0 + 4 * v + 16 * u;
            x[v][u] = (r < p.K) ? e[(size_t)r * p.lda] : 0.f;
          }
#pragma unroll
        for (int v = 0; v < 4; ++v)
#pragma unroll
          for (int u = 0; u < 8; ++u)
            if (r0 + 4 * v + 16 * u < p.K) acc[v] += x[v][u];
      }
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) lds[(q + 4 * v) * 64 + lane] = acc[v];
  }
  __syncthreads();
  KH_RT(12);
  const int c = tile * 64 + tid;
  if (tid < 64 && c < p.M) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += lds[i * 64 + tid];
    if (GRAD) {
      p.bias_io[c] = p.alpha * t;  // the bias gradient itself: the step follows the exchange between the ranks
    } else {
      float bc = p.alpha * t;
      if (p.mom != 0.f) bc = kh::mul_add(p.mom, p.bcorr[c], bc);
      if (p.bcorr) p.bcorr[c] = bc;
      p.bias_io[c] = -p.lr * bc + p.bias_io[c];
    }
  }
  KH_RT(13);
}

// Bias statistics of an RBM CD-1 step for 64 units, by one 256-thread workgroup at the end of the two-product
// weight launch (GemmParams::bias2_io): d = -g * colsum(neg) + mom * dbias ; d = g * colsum(pos) + d ; dbias = d ;
// bias += d.  Thread (grp = tid / 16, quad = tid % 16) sums rows grp, grp + 16, ... of 4 adjacent columns of the
// negative and of the positive matrix, 2 x 16 loads of 16 bytes in flight per 256 rows, added in row order; the 16
// partial sums are combined in the order 0..15 -- the sums of the stand-alone bias kernels, bit for bit.
__device__ __forceinline__ void rbm_bias_tile(const GemmParams &p, float *lds, int tile) {
  typedef float v4f __attribute__((ext_vector_type(4)));
  const int nh = (p.M + 63) >> 6;
  const bool hid = tile < nh;
  if (!hid) tile -= nh;
  const float *neg = hid ? p.A : p.B, *pos = hid ? p.A2 : p.B2;
  const int ld = hid ? p.lda : p.ldb, D = hid ? p.M : p.N, N = p.nk1 * kBK;
  float *bias = hid ? p.bias_io : p.bias2_io, *dbias = hid ? p.bcorr : p.bcorr2;
  const int grp = threadIdx.x >> 4, quad = threadIdx.x & 15;
  const int c0 = tile * 64 + quad * 4;
  v4f ap = {0.f, 0.f, 0.f, 0.f}, an = {0.f, 0.f, 0.f, 0.f};
  if (c0 < D) {  // (D is a multiple of 4: whole quads)
    for (int r0 = grp; r0 < N; r0 += 256) {
      v4f xp[16], xn[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int r = r0 + 16 * u;
        const v4f z = {0.f, 0.f, 0.f, 0.f};
        xp[u] = r < N ? *reinterpret_cast<const v4f *>(pos + (size_t)r * ld + c0) : z;
        xn[u] = r < N ? *reinterpret_cast<const v4f *>(neg + (size_t)r * ld + c0) : z;
      }
#pragma unroll
      for (int u = 0; u < 16; ++u)
        if (r0 + 16 * u < N) {
          ap += xp[u];
          an += xn[u];
        }
    }
  }
  v4f *pp = reinterpret_cast<v4f *>(lds), *pn = pp + 256;
  pp[grp * 16 + quad] = ap;
  pn[grp * 16 + quad] = an;
  __syncthreads();
  const int c = tile * 64 + threadIdx.x;
  if (threadIdx.x < 64 && c < D) {
    const float *fp = lds, *fn = lds + 1024;
    float tp = 0.f, tn = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      tp += fp[i * 64 + threadIdx.x];
      tn += fn[i * 64 + threadIdx.x];
    }
    float d = kh::mul_add(p.beta, dbias[c], p.alpha * tn);  // (alpha = -g, alpha2 = g, beta = momentum)
    d = kh::mul_add(p.alpha2, tp, d);
    dbias[c] = d;
    bias[c] = d + bias[c];
  }
}

// One tile (64 or 128 rows) of a weight update, and the bias half of the update where this workgroup carries it.
// The bias half rides in the update's launch: one workgroup of every tile row takes the column sums of the
// error for its outputs first (the columns it is about to stream as its A operand), 64 at a time, and then
// goes on to its tile: it runs that much behind the workgroups it shares the compute unit with and catches up
// when they are done, so the launch ends barely later than without the sums.  (Tried and measured: workgroups
// of their own in front of the tiles held 32 of the 1024 workgroup slots that the 64 x 64 tiles of a
// 2048 x 2048 layer fill exactly, and the 32 tiles left over started most of a workgroup lifetime late; at the
// end of the grid they start when the first tiles finish and add their whole 3.7 us -- kernel arguments, one
// round of loads, the exchange, the bias round trip -- to the launch.)  The chosen column walks with the row
// so that the carriers spread over the XCDs.
template <int BM, int BN, int BK, int WK, bool A_KMAJOR, bool B_KMAJOR, int EPI, bool VEC, int NW = 4>
__device__ __forceinline__ void update_tile(const GemmParams &p, float *lds, const int bid, const int ntiles) {
  static_assert(BM % 64 == 0, "tile rows in units of the 64 outputs of a bias tile");
  if (p.bias_tiles > 0) {
    int L = bid;  // same decode as gemm_tile
    if ((ntiles & 7) == 0) L = (L & 7) * (ntiles >> 3) + (L >> 3);
    int mi, ni, ks;
    if (p.n_fastest) {
      ni = L % p.tiles_n;
      mi = L / p.tiles_n;
      ks = 0;
    } else {
      mi = L % p.tiles_m;
      L /= p.tiles_m;
      ks = L % p.splitk;
      ni = L / p.splitk;
    }
    if (ks == 0 && ni == ((p.tiles_n >= 8) ? (mi & 7) * (p.tiles_n >> 3) : 0)) {
#pragma unroll
      for (int h = 0; h < BM / 64; ++h)
        if (mi * (BM / 64) + h < p.bias_tiles) {
          bias_update_tile<EPI == EPI_GRAD>(p, lds, mi * (BM / 64) + h);
          __syncthreads();
        }
    }
  }
  gemm_tile<BM, BN, BK, WK, A_KMAJOR, B_KMAJOR, EPI, VEC, NW>(p, lds, bid, ntiles);
}

template <int BM, int BN, int BK, int WK, bool A_KMAJOR, bool B_KMAJOR, int EPI, bool VEC, bool KTAIL = false>
__global__ __launch_bounds__(256) void gemm_kernel(GemmParams p) {
  __shared__ __attribute__((aligned(16))) float lds[gemm_lds_floats<BM, BN, BK, WK, A_KMAJOR, B_KMAJOR>()];
  if constexpr (EPI == EPI_UPDATE) {
    static_assert(gemm_lds_floats<BM, BN, BK, WK, A_KMAJOR, B_KMAJOR>() >= 16 * 64, "bias tile needs 1024 floats");
    // 64-row tiles: the bias half inside tile workgroups (update_tile); other tile heights: workgroups at the end
    const int ntiles = p.tiles_m * p.tiles_n * p.splitk;
    if constexpr (BM == 64) {
      update_tile<BM, BN, BK, WK, A_KMAJOR, B_KMAJOR, EPI, VEC>(p, lds, blockIdx.x, ntiles);
    } else if ((int)blockIdx.x >= ntiles) {
      bias_update_tile(p, lds, (int)blockIdx.x - ntiles);
    } else {
      gemm_tile<BM, BN, BK, WK, A_KMAJOR, B_KMAJOR, EPI, VEC>(p, lds, blockIdx.x, ntiles);
    }
  } else if constexpr (EPI == EPI_RBM) {
    static_assert(gemm_lds_floats<BM, BN, BK, WK, A_KMAJOR, B_KMAJOR>() >= 2048, "bias statistics need 2048 floats");
    const int ntiles = p.tiles_m * p.tiles_n * p.splitk;
    const int extra = (int)blockIdx.x - ntiles;
    if (extra < 0) {
      gemm_tile<BM, BN, BK, WK, A_KMAJOR, B_KMAJOR, EPI, VEC>(p, lds, blockIdx.x, ntiles);
    } else if (extra < p.bias_tiles) {
      rbm_bias_tile(p, lds, extra);
    } else {
      const int r = extra - p.bias_tiles;  // one row of the reconstruction error
      kh::mse_row_and_total(p.B + (size_t)r * p.ldb, p.B2 + (size_t)r * p.ldb, nullptr, p.N, p.mse_row_sq, r, p.mse_rows,
                            p.mse_ticket, p.mse_loss, lds, reinterpret_cast<double *>(lds + 32),
                            reinterpret_cast<unsigned *>(lds + 16));
    }
  } else {
    gemm_tile<BM, BN, BK, WK, A_KMAJOR, B_KMAJOR, EPI, VEC, 4, KTAIL>(p, lds, blockIdx.x, gridDim.x);
  }
}

#ifdef KHG_DEFINE_UPDATE_GROUP
// Up to kMaxGroup weight updates in one launch (GemmGroup above).  Measured on the eight updates of a
// Switchboard-sized step (tools/group_probe.py, tools/group_stamps.py): with 64 x 64 tiles 235 us against 262 us
// launched one by one.  In the long grid the workgroups of a compute unit drift out of phase by themselves
// (lifetimes 11-17 us), a start delay for half of the first round changed nothing beyond noise, and persistent
// workgroups looping over tiles (no re-dispatch) measured the same as one workgroup per tile.
// What bounds the 64 x 64 tile there is the compute unit's vector memory path, not the matrix cores: at K = 256 a
// tile loads 128 KB of operands (+ 16 KB of old weights, 16 KB stored) for 2.1 MFLOP, ~19 bytes per cycle and
// compute unit at the measured rate, and the four store instructions of a wave's epilogue took 4000 cycles to
// issue behind that traffic.  BIG: 128 x 128 tiles by 512-thread workgroups (8 waves of 64 x 32, two workgroups
// per compute unit = the same 4 waves per SIMD): half the operand bytes per FLOP.
template <bool BIG>
__global__ __launch_bounds__(BIG ? 512 : 256) __attribute__((amdgpu_waves_per_eu(4, 4))) void gemm_update_group_kernel(GemmGroup g) {
  constexpr int BT = BIG ? 128 : 64;
  __shared__ __attribute__((aligned(16))) float lds[gemm_lds_floats<BT, BT, kBK, 1, false, false>()];
  const int b = blockIdx.x;
  int idx = 0;
#pragma unroll
  for (int i = 1; i < kMaxGroup; ++i)
    if (i < g.n && b >= g.first[i]) idx = i;
  const GemmParams &p = g.p[idx];
  const int bid = b - g.first[idx];
  const int ntiles = p.tiles_m * p.tiles_n * p.splitk;
  if (bid >= ntiles) return;
  update_tile<BT, BT, kBK, 1, false, false, EPI_UPDATE, true, BIG ? 8 : 4>(p, lds, bid, ntiles);
}

void launch_update_group(hipStream_t st, const GemmGroup &g, bool big) {
  if (big) KHG_LAUNCH(gemm_update_group_kernel<true>, dim3(g.first[g.n]), dim3(512), 0, st, g);
  else KHG_LAUNCH(gemm_update_group_kernel<false>, dim3(g.first[g.n]), dim3(256), 0, st, g);
}
#endif  // KHG_DEFINE_UPDATE_GROUP

#ifdef KHG_DEFINE_CHAIN
// Several dependent forward products (hidden layers: bias + sigmoid epilogue) in ONE launch, one workgroup per
// compute unit and tile, the same tile in every layer.  A workgroup's tile of layer l needs the whole rows of layer
// l - 1's output for its row block, i.e. the tiles of the tiles_n workgroups that share its row block: each of them
// stores its tile write-through, waits for the stores, and adds 1 to the row block's counter of that layer; a
// consumer polls that counter (one lane, past the L1) until all tiles_n have arrived, and loads the activations
// past the L1.  Counters only grow: launch number `epoch` waits for tiles_n * epoch.  The grid (one tile each,
// at most one round of co-resident workgroups) is checked by the host; a poll that runs out of patience sets *fail
// and goes on, so the launch always ends.

template <int BM, int BN, int WK>
__global__ __launch_bounds__(256) void gemm_chain_kernel(GemmChain g) {
  __shared__ __attribute__((aligned(16))) float lds[gemm_lds_floats<BM, BN, kBK, WK, true, true>()];
  const int nblk = gridDim.x;
  int L = blockIdx.x;  // same decode as gemm_tile (n_fastest == 0, no K split over workgroups)
  if ((nblk & 7) == 0) L = (L & 7) * (nblk >> 3) + (L >> 3);
  const int tiles_m = g.p[0].tiles_m, tiles_n = g.p[0].tiles_n;
  const int mi = L % tiles_m;
  for (int l = 0; l < g.n; ++l) {
    if (l > 0) {
      if (threadIdx.x == 0) {
        const unsigned want = (unsigned)tiles_n * g.epoch;
        const unsigned *ctr = g.counters + (l - 1) * tiles_m + mi;
        int patience = 1 << 22;
        while ((int)(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) < 0) {
          __builtin_amdgcn_s_sleep(2);
          if (--patience == 0) {
            *g.fail = l;
            break;
          }
        }
      }
      __syncthreads();
    }
    gemm_tile<BM, BN, kBK, WK, true, true, EPI_BIAS_ACT, true, 4, false, true>(g.p[l], lds, blockIdx.x, nblk);
    if (l + 1 < g.n) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's write-through stores have reached the L2
      __syncthreads();
      if (threadIdx.x == 0) __hip_atomic_fetch_add(g.counters + l * tiles_m + mi, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

void launch_forward_chain(hipStream_t st, const GemmChain &g, int cfg) {
  const int nblk = g.p[0].tiles_m * g.p[0].tiles_n;
  if (cfg == 4) KHG_LAUNCH((gemm_chain_kernel<32, 64, 2>), dim3(nblk), dim3(256), 0, st, g);
  else KHG_LAUNCH((gemm_chain_kernel<32, 32, 4>), dim3(nblk), dim3(256), 0, st, g);
}
#endif  // KHG_DEFINE_CHAIN

#ifdef KHG_DEFINE_GRAD_GROUP
// The same grouped launch for the data-parallel form of the step: every problem writes its raw product
// e^T * in (and the column sums of e behind it, GemmParams::bias_io) instead of stepping the weights.
template <bool BIG>
__global__ __launch_bounds__(BIG ? 512 : 256) __attribute__((amdgpu_waves_per_eu(4, 4))) void gemm_grad_group_kernel(GemmGroup g) {
  constexpr int BT = BIG ? 128 : 64;
  __shared__ __attribute__((aligned(16))) float lds[gemm_lds_floats<BT, BT, kBK, 1, false, false>()];
  const int b = blockIdx.x;
  int idx = 0;
#pragma unroll
  for (int i = 1; i < kMaxGroup; ++i)
    if (i < g.n && b >= g.first[i]) idx = i;
  const GemmParams &p = g.p[idx];
  const int bid = b - g.first[idx];
  const int ntiles = p.tiles_m * p.tiles_n * p.splitk;
  if (bid >= ntiles) return;
  update_tile<BT, BT, kBK, 1, false, false, EPI_GRAD, true, BIG ? 8 : 4>(p, lds, bid, ntiles);
}

void launch_grad_group(hipStream_t st, const GemmGroup &g, bool big, hipEvent_t done) {
  // An event recorded behind the launch is a packet of its own in the stream's queue, and the launch after it waits
  // ~6 us for it (kernel trace of the data-parallel step); bound to the launch itself (hipExtLaunchKernelGGL's stop
  // event) the event is the kernel's own completion signal and the queue holds nothing extra.
  // (a launch that is being timed carries the timing's pair instead; its caller then marks `done` on its own)
  hipEvent_t start = done ? nullptr : g_bind_start, stop = done ? done : g_bind_stop;
  if (big) hipExtLaunchKernelGGL(gemm_grad_group_kernel<true>, dim3(g.first[g.n]), dim3(512), 0, st, start, stop, 0, g);
  else hipExtLaunchKernelGGL(gemm_grad_group_kernel<false>, dim3(g.first[g.n]), dim3(256), 0, st, start, stop, 0, g);
}
#endif  // KHG_DEFINE_GRAD_GROUP

// ---- sum of split-K slabs + the element-wise step that follows the GEMM ------------------------
template <int EPI, int VEC>
__global__ __launch_bounds__(256) void slab_reduce_kernel(GemmParams p, const float *slabs, int ld_slab) {
  const int c = (blockIdx.x * 256 + threadIdx.x) * VEC;
  if (c >= p.N) return;
  for (int r = blockIdx.y; r < p.M; r += gridDim.y) {
    float acc[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
    const float *src = slabs + (size_t)r * ld_slab + c;
    for (int s = 0; s < p.splitk; ++s) {
      if (VEC == 4) {
        const float4 v = *reinterpret_cast<const float4 *>(src + (size_t)s * p.slab_stride);
        acc[0] += v.x; acc[1] += v.y; acc[2] += v.z; acc[3] += v.w;
      } else {
        acc[0] += src[(size_t)s * p.slab_stride];
      }
    }
#pragma unroll
    for (int i = 0; i < VEC; ++i)
      if (c + i < p.N) epi_store<EPI>(p, r, c + i, epi_value<EPI>(p, acc[i], epi_aux<EPI>(p, r, c + i)), 0);
  }
}

template <int BM, int BN, int WK, bool AK, bool BK_, int EPI>
void launch_cfg(hipStream_t st, const GemmParams &p, bool vec) {
  const int nblk = p.tiles_m * p.tiles_n * p.splitk + ((EPI == EPI_UPDATE && BM != 64) || EPI == EPI_RBM ? p.bias_tiles : 0) +
                   (EPI == EPI_RBM ? p.mse_rows : 0);
  // K ends inside a tile: the variant that runs the tail in the fast loop (forward and back-propagation products on
  // the tiles whose waves split K -- the shapes of a 256-row bunch; K = 440, 1320, 9000 of the RM- and
  // Switchboard-sized nets)
  if constexpr ((EPI == EPI_BIAS_ACT || EPI == EPI_DSIG || EPI == EPI_BIAS_SAMPLE) && WK > 1 && KH_SCHED >= 2) {
    if (vec && p.splitk == 1 && p.a_bytes > 0 && p.K % (kBK * WK) != 0 && (((AK || BK_) ? (p.K & 3) : 0) == 0)) {
      KHG_LAUNCH((gemm_kernel<BM, BN, kBK, WK, AK, BK_, EPI, true, true>), dim3(nblk), dim3(256), p.dyn_lds, st, p);
      return;
    }
  }
  if (vec)
    KHG_LAUNCH((gemm_kernel<BM, BN, kBK, WK, AK, BK_, EPI, true>), dim3(nblk), dim3(256), p.dyn_lds, st, p);
  else
    KHG_LAUNCH((gemm_kernel<BM, BN, kBK, WK, AK, BK_, EPI, false>), dim3(nblk), dim3(256), p.dyn_lds, st, p);
}

template <bool AK, bool BK_, int EPI>
void launch_tiles(hipStream_t st, const GemmParams &p, int cfg, bool vec) {
  constexpr bool kWaveK = (EPI != EPI_UPDATE && EPI != EPI_RBM);  // the planner never picks 4, 5 for those
  switch (cfg) {
    case 0: launch_cfg<64, 64, 1, AK, BK_, EPI>(st, p, vec); break;
    case 1: launch_cfg<128, 64, 1, AK, BK_, EPI>(st, p, vec); break;
    case 2: launch_cfg<64, 128, 1, AK, BK_, EPI>(st, p, vec); break;
    case 3: launch_cfg<128, 128, 1, AK, BK_, EPI>(st, p, vec); break;
    case 4:
      if constexpr (kWaveK) launch_cfg<32, 64, 2, AK, BK_, EPI>(st, p, vec);
      break;
    default:
      if constexpr (kWaveK) launch_cfg<32, 32, 4, AK, BK_, EPI>(st, p, vec);
      break;
  }
}

template <int EPI>
void launch_layout(hipStream_t st, const GemmParams &p, int cfg, bool vec, bool a_kmajor, bool b_kmajor) {
  if (a_kmajor && b_kmajor) launch_tiles<true, true, EPI>(st, p, cfg, vec);
  else if (a_kmajor && !b_kmajor) launch_tiles<true, false, EPI>(st, p, cfg, vec);
  else if (!a_kmajor && b_kmajor) launch_tiles<false, true, EPI>(st, p, cfg, vec);
  else launch_tiles<false, false, EPI>(st, p, cfg, vec);
}

template <int EPI>
void launch_reduce(hipStream_t st, const GemmParams &p, const float *slabs, int ld_slab) {
  const bool vec = (ld_slab % 4 == 0) && (p.slab_stride % 4 == 0);
  const int per = vec ? 4 : 1;
  int gx = (p.N + 256 * per - 1) / (256 * per);
  int gy = p.M;
  const int cap = 8192 / (gx > 0 ? gx : 1);
  if (gy > cap) gy = cap < 1 ? 1 : cap;
  if (vec)
    hipLaunchKernelGGL((slab_reduce_kernel<EPI, 4>), dim3(gx, gy), dim3(256), 0, st, p, slabs, ld_slab);
  else
    hipLaunchKernelGGL((slab_reduce_kernel<EPI, 1>), dim3(gx, gy), dim3(256), 0, st, p, slabs, ld_slab);
}


#endif  // KHG_DEFINE_KERNELS

}  // namespace khg
#endif  // KALDI_AMD_CSRC_HIP_GEMM_KERNEL_H_
