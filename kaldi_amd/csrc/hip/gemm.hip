// fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32) with LDS-staged tiles, and the
// fused affine forward / backward / update entry points built on it.  Replaces the single
// cublasSgemm call site of the reference (src/cudamatrix/cu-matrix-inl.h:860-907) and the
// element-wise passes the reference ran around it (nnet/nnet-biasedlinearity.h:58-102,
// nnet/nnet-activation.h:207-218, nnet/nnet-rbm.h:136-223).
//
// Shape of the problem on this path: M = minibatch (256) for forward/backward, so a plain
// one-64x64-tile-per-workgroup decomposition leaves most of the 256 CUs idle.  Two ways out, both
// here: small tiles (32x64, 32x32) whose four waves split K among themselves and add their
// accumulators through LDS before the epilogue (a 256 x 2048 product is then 256 workgroups and
// finishes its elements -- bias + sigmoid, y(1-y) -- in the same kernel); and K split over
// workgroups, each writing an fp32 partial slab that the element-wise kernel which has to run
// next anyway (softmax, y(1-y) scaling, alpha/beta) sums in a fixed order.  No atomics either
// way: results are bit-reproducible.
//
// Kernel structure (one workgroup = 4 waves = 256 threads, one output tile BM x BN):
//   global --dwordx4--> VGPR (next K-tile, issued before the MFMAs of the current one)
//          --ds_write_b128--> LDS double buffer (one barrier per K-tile)
//          --ds_read_b128 / ds_read_b32--> MFMA operands, 32x32 accumulators in registers.
// Operand k-order inside a group of 8 is permuted (lane half h, step j uses k = 8g + 4h + j)
// so that a K-contiguous operand is fetched with one 16-byte LDS read per 4 MFMAs; both
// operands use the same permutation, so the sum is over the same k set.
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <type_traits>
#include <utility>
#include <vector>

#include "common.h"

namespace kh {
int softmax_slabs_launch(hipStream_t st, float *y, int ldy, const float *x, int ldx, int nslab, size_t slab_stride,
                         const float *bias, int rows, int cols, const XentArgs *xent);
}

#ifndef KH_SCHED
#define KH_SCHED 3  // K-loop form: 0 group-by-group fragments, 2 whole-tile fragments, 3 whole-tile except in the update GEMMs
#endif

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum Epi {
  EPI_AXPBY = 0,   // C = alpha*acc + beta*C
  EPI_SLAB = 1,    // slab[ks] = acc                     (split-K partial)
  EPI_BIAS_ACT = 2,// C = act(bias[col] + acc)
  EPI_DSIG = 3,    // C = acc * y * (1 - y)
  EPI_UPDATE = 4,  // corr = g*acc + mom*corr ; W = W - l2*W - lr*corr
  EPI_RBM = 5      // dW = g*acc + dW - lrl2*W ; W += dW
};

struct GemmParams {
  const float *A, *B;
  int lda, ldb;
  int M, N, K;
  int splitk, k_per_split;
  int a_bytes, b_bytes;  // operand footprints when both are below 2 GiB (buffer addressing), else 0
  int tiles_m, tiles_n;
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
#ifdef KH_GEMM_STAMPS
  unsigned long long *stamps;
#endif
};

#ifdef KH_GEMM_STAMPS
// diagnostic build only: cycle stamps of wave 0 of every workgroup, 16 slots per workgroup
#define KH_STAMP(i)                                                                     \
  do {                                                                                  \
    if (p.stamps && threadIdx.x == 0) p.stamps[(size_t)blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define KH_STAMP(i) do { } while (0)
#endif

// The epilogue's second operand of one output element, fetched before any store so that the
// loads of all elements of a lane are in flight together: the bias of the column, or y.  Row and
// column are clamped into the matrix (elements outside are computed but never stored).
template <int EPI>
__device__ __forceinline__ float epi_aux(const GemmParams &p, int row, int col) {
  if (EPI == EPI_BIAS_ACT) return p.bias[min(col, p.N - 1)];
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
  return acc;
}

template <int EPI>
__device__ __forceinline__ void epi_store(const GemmParams &p, int row, int col, float val, int ks) {
  if (EPI == EPI_SLAB) {
    p.C[(size_t)ks * p.slab_stride + (size_t)row * p.ldc + col] = val;
  } else if (EPI == EPI_AXPBY) {
    float *c = p.C + (size_t)row * p.ldc + col;
    *c = (p.beta == 0.f) ? p.alpha * val : p.alpha * val + p.beta * *c;
  } else if (EPI == EPI_BIAS_ACT || EPI == EPI_DSIG) {
    p.C[(size_t)row * p.ldc + col] = val;
  } else if (EPI == EPI_UPDATE) {
    float c = p.alpha * val;
    if (p.mom != 0.f) c += p.mom * p.C[(size_t)row * p.ldc + col];
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
  }
}

// BM x BN output tile, 4 waves arranged WAVES_M x WAVES_N x WK.  A K-tile is BKT = BK*WK deep and
// wave-slice wk of it (BK = 32 k) goes to the waves with that wk: with WK > 1 the waves of a
// workgroup split K among themselves and add their accumulators through LDS before the epilogue.
// That keeps 256 workgroups (one per CU) on a 256-row x 2048-column product with 32x64 tiles and
// the whole K inside the workgroup, so the epilogue can finish the element (bias, sigmoid, ...)
// instead of writing a partial slab for a second kernel.
// A_KMAJOR: A(m,k) at A[m*lda+k] (k contiguous), otherwise A[k*lda+m].  Same for B with n.
// VEC: every row base is 16-byte aligned and leading dimensions are multiples of 4 floats.
template <int BM, int BN, int BK, int WK, bool A_KMAJOR, bool B_KMAJOR, int EPI, bool VEC>
__global__ __launch_bounds__(256) void gemm_kernel(GemmParams p) {
  constexpr int NT = 256;
  constexpr int WAVES_M = (WK == 1) ? 2 : 1, WAVES_N = (WK == 4) ? 1 : 2;
  static_assert(WAVES_M * WAVES_N * WK == 4, "4 waves per workgroup");
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

  __shared__ __attribute__((aligned(16))) float lds[2 * (A_ELEMS + B_ELEMS)];

  // ---- which tile: XCD-aware decode of the linear block id -----------------------------------
  // Blocks b and b+8 run on the same XCD (same L2); give each XCD a contiguous run of tiles so
  // that neighbours, which share the B (weight) panel, hit in that L2.
  KH_STAMP(0);
  const int nblk = gridDim.x;
  int L = blockIdx.x;
  if ((nblk & 7) == 0) L = (L & 7) * (nblk >> 3) + (L >> 3);
  const int mi = L % p.tiles_m;
  L /= p.tiles_m;
  const int ks = L % p.splitk;
  const int ni = L / p.splitk;
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
  auto load_fast = [&](int kt) {
#pragma unroll
    for (int u = 0; u < A_UNITS; ++u) ra[u] = buf_load(rsrc_a, oa[u], kt * a_step);
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
    const float4 v = (c < A_UNITS) ? buf_load(rsrc_a, oa[c < A_UNITS ? c : 0], kt * a_step)
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
  constexpr bool PREFETCH = (EPI == EPI_UPDATE);
  static_assert(!(PREFETCH && WK > 1), "the update epilogue is not built for K split over waves");
  constexpr int NPRE = PREFETCH ? TM * TN : 1;
  float pre[NPRE][16];
  const bool full_tile = (m0 + BM <= p.M) && (n0 + BN <= p.N);
  int npre = 0;  // accumulators whose operand is in `pre`
  auto prefetch_acc = [&](int a) {
    const int i = a / TN, q = a % TN;
    const float *src = p.W + (size_t)(m0 + wm0 + i * 32 + 4 * lh) * p.ldw + (n0 + wn0 + q * 32 + li);
#pragma unroll
    for (int r = 0; r < 16; ++r) pre[a][r] = src[(size_t)((r & 3) + 8 * (r >> 2)) * p.ldw];
  };

  // K-tiles completely inside the split take the fast loads; at most one tail tile does not.
  const int nfull = (VEC && p.a_bytes > 0) ? (kend - kbeg) / BKT : 0;
  constexpr int G = BK / 8;           // k-groups per tile
  // which K-loop: whole-tile operand fragments (more registers: 2 waves per SIMD at most) or
  // group-by-group fragments (the update GEMMs, which run 4 workgroups per CU)
  constexpr bool kWholeTile = (KH_SCHED == 2) || (KH_SCHED == 3 && EPI != EPI_UPDATE && EPI != EPI_RBM);
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
    load_fast(0);
    KH_STAMP(1);
    store_tiles(0);
    KH_STAMP(2);
    __syncthreads();
    KH_STAMP(3);
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
      if (PREFETCH && full_tile) {
#pragma unroll
        for (int a = 0; a < NPRE; ++a)
          if (kt == a) {
            prefetch_acc(a);
            npre = a + 1;
          }
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
      if (PREFETCH && full_tile) {
#pragma unroll
        for (int a = 0; a < NPRE; ++a)
          if (kt == a) {
            prefetch_acc(a);
            npre = a + 1;
          }
      }
      tile_body(std::false_type());
    };
    for (; kt + 1 < nfull; ++kt) steady();
    KH_STAMP(4);
    const bool tail = nkt > nfull;  // the K tail, at most one partial tile
    if (tail) load_slow(nfull);
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
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int q = 0; q < TN; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          const int col = n0 + wn0 + q * 32 + li;
          if (PREFETCH && i * TN + q < npre) {
            // full tile, old weight already here (same arithmetic as epi_store<EPI_UPDATE>)
            float c = p.alpha * acc[i][q][r];
            if (p.mom != 0.f) c += p.mom * p.C[(size_t)row * p.ldc + col];
            if (p.C) p.C[(size_t)row * p.ldc + col] = c;
            float wv = pre[PREFETCH ? i * TN + q : 0][r];
            if (p.l2 != 0.f) wv = -p.l2 * wv + wv;
            p.Wout[(size_t)row * p.ldwout + col] = -p.lr * c + wv;
          } else if (row < p.M && col < p.N) {
            epi_store<EPI>(p, row, col, epi_value<EPI>(p, acc[i][q][r], epi_aux<EPI>(p, row, col)), ks);
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
        if (row < p.M && col < p.N) epi_store<EPI>(p, row, col, sum[a][rr], ks);
      }
  }
  KH_STAMP(6);
#ifdef KH_GEMM_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  KH_STAMP(7);
#endif
}

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

// ---- host side ---------------------------------------------------------------------------------

struct TileCfg {
  int bm, bn;
  int wk;          // waves that split K inside the workgroup (1 = none)
  int lds_bytes;   // LDS per workgroup (K-major operands, the larger image)
  int vgprs;       // registers per lane, rounded
  double fixed;    // prologue + last tile + epilogue of a workgroup, cycles (measured with in-kernel stamps)
};
const TileCfg kCfgs[] = {{64, 64, 1, 36864, 104, 4400.0},  {128, 64, 1, 55296, 176, 5500.0}, {64, 128, 1, 55296, 176, 5500.0},
                         {128, 128, 1, 73728, 208, 8300.0}, {32, 64, 2, 52224, 128, 4400.0},  {32, 32, 4, 67584, 128, 4400.0}};
constexpr int kBK = 32;

struct Plan {
  int cfg;     // index into kCfgs
  int splitk;
  int k_per_split;
};

// Picks the tile and the K split with a small cycle model of the kernel, checked against sweeps over
// every tile / split for the Switchboard- and RM-sized layers (tools/gemm_sweep.py).  Candidates: the
// 64x64 tile (within 2 % of the best of the large tiles everywhere, clearly best for the
// weight-update GEMMs; the larger ones only when forced) and, where the epilogue allows it, the
// two tiles whose waves split K.  Model: the busiest CU runs ceil(workgroups / CUs) workgroups one
// after another; a workgroup costs its MFMAs x 1.28 (what the 64x64 kernel measures against the
// bare MFMA count; for that tile written as the ~12.5 B/clk of operand staging a CU sustained in
// the first version of the kernel); its fixed part (prologue, last tile, epilogue: ~4400 cycles by
// in-kernel stamps) hides behind the K loops of the workgroups that share the CU, so it is exposed
// once per round of co-resident workgroups; a split K adds the slab round trip and the launch of
// the kernel that sums the slabs.
int force_bm = -1, force_bn = -1, force_s = -1;
bool env_read = false;

Plan make_plan(int M, int N, int K, int num_cu, bool allow_split, bool always_slabs, bool allow_wk) {
  static bool wk_on = true;
  if (!env_read) {
    env_read = true;
    const char *e = getenv("KALDI_HIP_GEMM_FORCE");  // "BM,BN,S" for experiments
    if (e) sscanf(e, "%d,%d,%d", &force_bm, &force_bn, &force_s);
    const char *w = getenv("KALDI_HIP_GEMM_WK");      // "0": never split K over the waves of a workgroup
    if (w) wk_on = atoi(w) != 0;
  }
  Plan best = {0, 1, ((K + kBK - 1) / kBK) * kBK};
  double best_cost = 1e300;
  const int ncfg = (int)(sizeof(kCfgs) / sizeof(kCfgs[0]));
  const double cyc_per_us = 2050.0;  // clock held under fp32 MFMA load
  for (int c = 0; c < ncfg; ++c) {
    const TileCfg &t = kCfgs[c];
    if (t.wk > 1 && !allow_wk) continue;
    if (force_bm > 0 ? (t.bm != force_bm || t.bn != force_bn) : !(c == 0 || (t.wk > 1 && wk_on))) continue;
    const int bkt = kBK * t.wk;
    const int ktiles = (K + bkt - 1) / bkt;
    const int tm = (M + t.bm - 1) / t.bm, tn = (N + t.bn - 1) / t.bn;
    for (int s = 1; s <= 16; s *= 2) {
      if (s > 1 && !allow_split) break;
      if (force_s > 0 && s != force_s) continue;
      if (s > ktiles) break;
      const int kt_per = (ktiles + s - 1) / s;
      const int s_eff = (ktiles + kt_per - 1) / kt_per;
      if (s_eff != s) continue;
      const double blocks = (double)tm * tn * s;
      const double kper = (double)kt_per * bkt;
      const double mfma = (double)t.bm * t.bn * kper * 2.0 / 256.0 * 1.08;
      // (K split over waves: same MFMA count per byte staged as the 64x64 tile is charged below)
      const double mem = t.wk > 1 ? mfma * (1.28 / 1.08) : (double)(t.bm + t.bn) * kper * 4.0 / 12.5;
      const double per_block = mfma > mem ? mfma : mem;
      const double n = ceil(blocks / num_cu);
      // with two or more workgroups on every CU one's stalls are filled by another's MFMAs
      static const double share_f = getenv("KALDI_HIP_GEMM_SHARE") ? atof(getenv("KALDI_HIP_GEMM_SHARE")) : 1.0;
      const double share = blocks >= 2.0 * num_cu ? share_f : 1.0;
      // The fixed part of a workgroup (prologue, last tile, epilogue) hides behind the K loops of the
      // workgroups that share its CU, but only one of every `co` (how many fit a CU) stays exposed:
      // short-K products with many small tiles (the data-parallel gradient GEMMs) pay it many times.
      const double co = floor(163840.0 / t.lds_bytes) < floor(512.0 / t.vgprs) ? floor(163840.0 / t.lds_bytes)
                                                                                 : floor(512.0 / t.vgprs);
      const double rounds = n / (co > 1.0 ? co : 1.0);
      double cost = (n * per_block * share + (rounds > 1.0 ? rounds : 1.0) * t.fixed) / cyc_per_us;
      if (s > 1 || always_slabs) cost += (s + 1.0) * (double)M * N * 4.0 / 3.5e6 + 2.5;
      if (cost < best_cost) {
        best_cost = cost;
        best.cfg = c;
        best.splitk = s;
        best.k_per_split = kt_per * bkt;
      }
    }
  }
  return best;
}

template <int BM, int BN, int WK, bool AK, bool BK_, int EPI>
void launch_cfg(hipStream_t st, const GemmParams &p, bool vec) {
  const int nblk = p.tiles_m * p.tiles_n * p.splitk;
  if (vec)
    hipLaunchKernelGGL((gemm_kernel<BM, BN, kBK, WK, AK, BK_, EPI, true>), dim3(nblk), dim3(256), 0, st, p);
  else
    hipLaunchKernelGGL((gemm_kernel<BM, BN, kBK, WK, AK, BK_, EPI, false>), dim3(nblk), dim3(256), 0, st, p);
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

// ---- optional per-launch timing -------------------------------------------------------------------
struct GemmTiming {
  bool enabled = false;
  std::vector<hipEvent_t> pool;                          // recycled events
  std::vector<std::pair<hipEvent_t, hipEvent_t> > spans;  // one per launch since the last read
  double flops = 0.0;
};
GemmTiming &timing() {
  static GemmTiming t;
  return t;
}
hipEvent_t timing_event() {
  GemmTiming &t = timing();
  hipEvent_t e;
  if (!t.pool.empty()) {
    e = t.pool.back();
    t.pool.pop_back();
  } else if (hipEventCreate(&e) != hipSuccess) {
    return nullptr;
  }
  return e;
}
struct TimedLaunch {
  hipEvent_t e0 = nullptr, e1 = nullptr;
  hipStream_t st;
  TimedLaunch(hipStream_t s, double fl) : st(s) {
    if (!timing().enabled) return;
    e0 = timing_event();
    e1 = timing_event();
    if (e0 && e1) {
      (void)hipEventRecord(e0, st);
      timing().flops += fl;
    }
  }
  ~TimedLaunch() {
    if (e0 && e1) {
      (void)hipEventRecord(e1, st);
      timing().spans.push_back(std::make_pair(e0, e1));
    }
  }
};

// Runs op(A)*op(B) and the epilogue EPI.  `softmax` selects the row-softmax finish (always via
// slabs, since a row spans many tiles).
template <int EPI>
int run_gemm(khStream_t s, bool a_kmajor, bool b_kmajor, GemmParams p, bool allow_split, bool softmax = false,
             const kh::XentArgs *xent = nullptr) {
  hipStream_t st = kh::pick(s);
  const bool vec = kh::aligned16(p.A) && kh::aligned16(p.B) && (p.lda % 4 == 0) && (p.ldb % 4 == 0);
  Plan plan = make_plan(p.M, p.N, p.K, kh::ctx().num_cu > 0 ? kh::ctx().num_cu : 256, allow_split || softmax,
                        softmax, EPI != EPI_UPDATE && EPI != EPI_RBM);
  const int bm = kCfgs[plan.cfg].bm, bn = kCfgs[plan.cfg].bn;
  p.tiles_m = (p.M + bm - 1) / bm;
  p.tiles_n = (p.N + bn - 1) / bn;
  p.splitk = plan.splitk;
  p.k_per_split = plan.k_per_split;
  {
    // rows x leading dimension of each operand as stored; buffer addressing needs < 2 GiB
    const double ab = ((double)((a_kmajor ? p.M : p.K) - 1) * p.lda + (a_kmajor ? p.K : p.M)) * 4.0;
    const double bb = ((double)((b_kmajor ? p.N : p.K) - 1) * p.ldb + (b_kmajor ? p.K : p.N)) * 4.0;
    const bool ok = ab < 2147483648.0 && bb < 2147483648.0 && p.K > 0;
    p.a_bytes = ok ? (int)ab : 0;
    p.b_bytes = ok ? (int)bb : 0;
  }
  const double flops = 2.0 * p.M * p.N * p.K;
  if (p.splitk == 1 && !softmax) {
    TimedLaunch tl(st, flops);
    launch_layout<EPI>(st, p, plan.cfg, vec, a_kmajor, b_kmajor);
    return kh::launch_status();
  }
  // partial slabs in the workspace, then the fused finish
  const int ld_slab = (p.N + 3) & ~3;
  const size_t slab_stride = (size_t)p.M * ld_slab;
  void *ws;
  // (+ 8 bytes per row: per-row results of the fused softmax / cross-entropy finish)
  int rc = kh::ensure_workspace(st, slab_stride * p.splitk * sizeof(float) + (xent ? (size_t)p.M * 8 : 0), &ws);
  if (rc != KH_OK) return rc;
  GemmParams ps = p;
  ps.C = (float *)ws;
  ps.ldc = ld_slab;
  ps.slab_stride = slab_stride;
  {
    TimedLaunch tl(st, flops);
    launch_layout<EPI_SLAB>(st, ps, plan.cfg, vec, a_kmajor, b_kmajor);
  }
  rc = kh::launch_status();
  if (rc != KH_OK) return rc;
  p.slab_stride = slab_stride;
  if (softmax)
    return kh::softmax_slabs_launch(st, p.C, p.ldc, (const float *)ws, ld_slab, p.splitk, slab_stride, p.bias, p.M,
                                    p.N, xent);
  launch_reduce<EPI>(st, p, (const float *)ws, ld_slab);
  return kh::launch_status();
}

#ifdef KH_GEMM_STAMPS
unsigned long long *g_stamps = nullptr;
constexpr size_t kStampSlots = 16 * 65536;
#endif

GemmParams base_params(int M, int N, int K, const float *A, int lda, const float *B, int ldb, float *C, int ldc) {
  GemmParams p;
  memset(&p, 0, sizeof(p));
  p.A = A; p.B = B; p.lda = lda; p.ldb = ldb;
  p.M = M; p.N = N; p.K = K;
  p.C = C; p.ldc = ldc;
  p.alpha = 1.f; p.beta = 0.f;
#ifdef KH_GEMM_STAMPS
  if (!g_stamps && hipMalloc((void **)&g_stamps, kStampSlots * 8) != hipSuccess) g_stamps = nullptr;
  p.stamps = g_stamps;
#endif
  return p;
}

// vectors of the bias update: bcorr = g*colsum(e) + mom*bcorr ; b -= lr*bcorr
__global__ __launch_bounds__(1024) void bias_update_kernel(const float *e, int lde, int N, int O, float g, float mom,
                                                           float lr, float *bias, float *bcorr) {
  __shared__ float part[16][64];
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  float acc = 0.f;
  if (c < O)
    for (int r = grp; r < N; r += 16) acc += e[(size_t)r * lde + c];
  part[grp][lane] = acc;
  __syncthreads();
  if (grp == 0 && c < O) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += part[i][lane];
    float bc = g * t;
    if (mom != 0.f) bc += mom * bcorr[c];
    if (bcorr) bcorr[c] = bc;
    bias[c] = -lr * bc + bias[c];
  }
}

// RBM bias statistics: d = mom*d - g*colsum(neg) + g*colsum(pos) ; b += d
__global__ __launch_bounds__(1024) void rbm_bias_kernel(const float *pos, int ldp, const float *neg, int ldn, int N,
                                                        int D, float g, float mom, float *bias, float *dbias) {
  __shared__ float pp[16][64];
  __shared__ float pn[16][64];
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  float ap = 0.f, an = 0.f;
  if (c < D)
    for (int r = grp; r < N; r += 16) {
      ap += pos[(size_t)r * ldp + c];
      an += neg[(size_t)r * ldn + c];
    }
  pp[grp][lane] = ap;
  pn[grp][lane] = an;
  __syncthreads();
  if (grp == 0 && c < D) {
    float tp = 0.f, tn = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      tp += pp[i][lane];
      tn += pn[i][lane];
    }
    float d = -g * tn + mom * dbias[c];
    d = g * tp + d;
    dbias[c] = d;
    bias[c] = d + bias[c];
  }
}

// element-wise second half of the data-parallel update (same formulas as the EPI_UPDATE epilogue)
template <int VEC>
__global__ __launch_bounds__(256) void affine_apply_kernel(int O, int I, const float *grad, int ldg, float g,
                                                           float mom, float lr, float l2, float *W, int ldw,
                                                           float *corr, int ldcorr) {
  const int c = (blockIdx.x * 256 + threadIdx.x) * VEC;
  if (c >= I) return;
  for (int r = blockIdx.y; r < O; r += gridDim.y) {
    float gv[VEC], cv[VEC], wv[VEC];
    if (VEC == 4) {
      *reinterpret_cast<float4 *>(gv) = *reinterpret_cast<const float4 *>(grad + (size_t)r * ldg + c);
      *reinterpret_cast<float4 *>(wv) = *reinterpret_cast<const float4 *>(W + (size_t)r * ldw + c);
      if (mom != 0.f) *reinterpret_cast<float4 *>(cv) = *reinterpret_cast<const float4 *>(corr + (size_t)r * ldcorr + c);
    } else {
      gv[0] = grad[(size_t)r * ldg + c];
      wv[0] = W[(size_t)r * ldw + c];
      if (mom != 0.f) cv[0] = corr[(size_t)r * ldcorr + c];
    }
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      float cc = g * gv[i];
      if (mom != 0.f) cc += mom * cv[i];
      cv[i] = cc;
      float w = wv[i];
      if (l2 != 0.f) w = -l2 * w + w;
      wv[i] = -lr * cc + w;
    }
    if (VEC == 4) {
      *reinterpret_cast<float4 *>(W + (size_t)r * ldw + c) = *reinterpret_cast<float4 *>(wv);
      if (corr) *reinterpret_cast<float4 *>(corr + (size_t)r * ldcorr + c) = *reinterpret_cast<float4 *>(cv);
    } else {
      W[(size_t)r * ldw + c] = wv[0];
      if (corr) corr[(size_t)r * ldcorr + c] = cv[0];
    }
  }
}

__global__ void bias_apply_kernel(int O, const float *bgrad, float g, float mom, float lr, float *bias, float *bcorr) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= O) return;
  float bc = g * bgrad[c];
  if (mom != 0.f) bc += mom * bcorr[c];
  if (bcorr) bcorr[c] = bc;
  bias[c] = -lr * bc + bias[c];
}

}  // namespace

KH_EXPORT int hipF_affine_apply(khStream_t s, int32_t O, int32_t I, const float *grad, int32_t ldg, const float *bgrad,
                                float gscale, float momentum, float lr, float l2, float *W, int32_t ldw, float *corr,
                                int32_t ldcorr, float *bias, float *bcorr) {
  KH_CHECK_INIT();
  KH_REQUIRE(O > 0 && I > 0 && grad && bgrad && W && bias && ldg >= I && ldw >= I);
  KH_REQUIRE(momentum == 0.f || (corr != nullptr && bcorr != nullptr));
  KH_REQUIRE(corr == nullptr || ldcorr >= I);
  const bool vec = (I % 4 == 0) && (ldg % 4 == 0) && (ldw % 4 == 0) && (!corr || ldcorr % 4 == 0) &&
                   kh::aligned16(grad) && kh::aligned16(W) && (!corr || kh::aligned16(corr));
  const int per = vec ? 4 : 1;
  int gx = (I + 256 * per - 1) / (256 * per), gy = O;
  const int cap = 8192 / gx;
  if (gy > cap) gy = cap < 1 ? 1 : cap;
  hipStream_t st = kh::pick(s);
  if (vec)
    hipLaunchKernelGGL((affine_apply_kernel<4>), dim3(gx, gy), dim3(256), 0, st, O, I, grad, ldg, gscale, momentum, lr,
                       l2, W, ldw, corr, ldcorr);
  else
    hipLaunchKernelGGL((affine_apply_kernel<1>), dim3(gx, gy), dim3(256), 0, st, O, I, grad, ldg, gscale, momentum, lr,
                       l2, W, ldw, corr, ldcorr);
  hipLaunchKernelGGL(bias_apply_kernel, dim3((O + 255) / 256), dim3(256), 0, st, O, bgrad, gscale, momentum, lr, bias,
                     bcorr);
  return kh::launch_status();
}

#ifdef KH_GEMM_STAMPS
// diagnostic build only: copies the stamps of the last GEMM launch (16 per workgroup) to the host
KH_EXPORT int hipK_gemm_stamps_read(unsigned long long *dst, size_t slots) {
  KH_REQUIRE(dst && slots <= kStampSlots && g_stamps);
  KH_HIP(hipDeviceSynchronize());
  KH_HIP(hipMemcpy(dst, g_stamps, slots * 8, hipMemcpyDeviceToHost));
  return KH_OK;
}
#endif

KH_EXPORT int hipK_gemm_force_plan(int bm, int bn, int splitk) {
  env_read = true;
  force_bm = bm > 0 ? bm : -1;
  force_bn = bn > 0 ? bn : -1;
  force_s = splitk > 0 ? splitk : -1;
  return KH_OK;
}

KH_EXPORT int hipK_gemm_timing(int enable) {
  timing().enabled = enable != 0;
  return KH_OK;
}

KH_EXPORT int hipK_gemm_timing_read(double *total_ms, double *busy_ms, long long *launches, double *flops) {
  GemmTiming &t = timing();
  double ms = 0.0;
  std::vector<std::pair<double, double> > iv;  // [start, end) of every launch, ms since the first one's start
  for (size_t i = 0; i < t.spans.size(); ++i) {
    KH_HIP(hipEventSynchronize(t.spans[i].second));
    float one = 0.f, off = 0.f;
    KH_HIP(hipEventElapsedTime(&one, t.spans[i].first, t.spans[i].second));
    if (i > 0) {
      KH_HIP(hipEventSynchronize(t.spans[i].first));
      KH_HIP(hipEventElapsedTime(&off, t.spans[0].first, t.spans[i].first));
    }
    ms += one;
    iv.push_back(std::make_pair((double)off, (double)off + one));
  }
  // length of the union of the intervals
  std::sort(iv.begin(), iv.end());
  double busy = 0.0, cur_a = 0.0, cur_b = -1.0;
  for (size_t i = 0; i < iv.size(); ++i) {
    if (cur_b < cur_a || iv[i].first > cur_b) {
      if (cur_b > cur_a) busy += cur_b - cur_a;
      cur_a = iv[i].first;
      cur_b = iv[i].second;
    } else if (iv[i].second > cur_b) {
      cur_b = iv[i].second;
    }
  }
  if (cur_b > cur_a) busy += cur_b - cur_a;
  for (size_t i = 0; i < t.spans.size(); ++i) {
    t.pool.push_back(t.spans[i].first);
    t.pool.push_back(t.spans[i].second);
  }
  if (total_ms) *total_ms = ms;
  if (busy_ms) *busy_ms = busy;
  if (launches) *launches = (long long)t.spans.size();
  if (flops) *flops = t.flops;
  t.spans.clear();
  t.flops = 0.0;
  return KH_OK;
}

KH_EXPORT int hipF_gemm(khStream_t s, int transA, int transB, int32_t M, int32_t N, int32_t K, float alpha,
                        const float *A, int32_t lda, const float *B, int32_t ldb, float beta, float *C, int32_t ldc) {
  KH_CHECK_INIT();
  KH_REQUIRE((transA == KH_NO_TRANS || transA == KH_TRANS) && (transB == KH_NO_TRANS || transB == KH_TRANS));
  KH_REQUIRE(M >= 0 && N >= 0 && K >= 0);
  if (M == 0 || N == 0) return KH_OK;
  KH_REQUIRE(C != nullptr && ldc >= N);
  const bool a_kmajor = (transA == KH_NO_TRANS), b_kmajor = (transB == KH_TRANS);
  if (K > 0) {
    KH_REQUIRE(A != nullptr && B != nullptr);
    KH_REQUIRE(lda >= (a_kmajor ? K : M) && ldb >= (b_kmajor ? K : N));
  }
  GemmParams p = base_params(M, N, K, A, lda, B, ldb, C, ldc);
  p.alpha = alpha;
  p.beta = beta;
  return run_gemm<EPI_AXPBY>(s, a_kmajor, b_kmajor, p, /*allow_split=*/true);
}

KH_EXPORT int hipF_affine_forward(khStream_t s, int act, int trans_w, int32_t N, int32_t O, int32_t I,
                                  const float *in, int32_t ldin, const float *W, int32_t ldw, const float *bias,
                                  float *out, int32_t ldout) {
  KH_CHECK_INIT();
  KH_REQUIRE(act == KH_ACT_NONE || act == KH_ACT_SIGMOID || act == KH_ACT_SOFTMAX);
  KH_REQUIRE(trans_w == KH_TRANS || trans_w == KH_NO_TRANS);
  KH_REQUIRE(N >= 0 && O > 0 && I > 0);
  if (N == 0) return KH_OK;
  KH_REQUIRE(in && W && bias && out && ldin >= I && ldout >= O);
  const bool b_kmajor = (trans_w == KH_TRANS);  // W is O x I (k contiguous) or I x O
  KH_REQUIRE(ldw >= (b_kmajor ? I : O));
  GemmParams p = base_params(N, O, I, in, ldin, W, ldw, out, ldout);
  p.bias = bias;
  p.act = act;
  return run_gemm<EPI_BIAS_ACT>(s, true, b_kmajor, p, true, act == KH_ACT_SOFTMAX);
}

KH_EXPORT int hipF_affine_forward_xent(khStream_t s, int trans_w, int32_t N, int32_t O, int32_t I, const float *in,
                                       int32_t ldin, const float *W, int32_t ldw, const float *bias, float *post,
                                       int32_t ldpost, int write_post, const int32_t *tgt, float *diff, int32_t lddiff,
                                       float *log_post, int32_t *max_id, long long *counters, double *loss) {
  KH_CHECK_INIT();
  KH_REQUIRE(trans_w == KH_TRANS || trans_w == KH_NO_TRANS);
  KH_REQUIRE(N >= 0 && O > 0 && I > 0);
  if (N == 0) return KH_OK;
  KH_REQUIRE(in && W && bias && post && ldin >= I && ldpost >= O);
  KH_REQUIRE(tgt && diff && diff != post && lddiff == ldpost && counters && loss);
  const bool b_kmajor = (trans_w == KH_TRANS);
  KH_REQUIRE(ldw >= (b_kmajor ? I : O));
  GemmParams p = base_params(N, O, I, in, ldin, W, ldw, post, ldpost);
  p.bias = bias;
  p.act = KH_ACT_SOFTMAX;
  kh::XentArgs xa;
  xa.tgt = tgt;
  xa.diff = diff;
  xa.lddiff = lddiff;
  xa.log_post = log_post;
  xa.max_id = max_id;
  xa.counters = counters;
  xa.loss = loss;
  xa.write_post = write_post;
  return run_gemm<EPI_BIAS_ACT>(s, true, b_kmajor, p, true, true, &xa);
}

KH_EXPORT int hipF_affine_backward(khStream_t s, int32_t N, int32_t O, int32_t I, const float *e, int32_t lde,
                                   const float *W, int32_t ldw, const float *y, int32_t ldy, float *eout,
                                   int32_t ldeout) {
  KH_CHECK_INIT();
  KH_REQUIRE(N >= 0 && O > 0 && I > 0);
  if (N == 0) return KH_OK;
  KH_REQUIRE(e && W && eout && lde >= O && ldw >= I && ldeout >= I && (y == nullptr || ldy >= I));
  GemmParams p = base_params(N, I, O, e, lde, W, ldw, eout, ldeout);
  if (y) {
    p.y = y;
    p.ldy = ldy;
    return run_gemm<EPI_DSIG>(s, true, false, p, true);
  }
  return run_gemm<EPI_AXPBY>(s, true, false, p, true);
}

KH_EXPORT int hipF_affine_update_weights(khStream_t s, int32_t N, int32_t O, int32_t I, const float *e, int32_t lde,
                                         const float *in, int32_t ldin, float gscale, float momentum, float lr,
                                         float l2, const float *W_in, int32_t ldw_in, float *W_out, int32_t ldw_out,
                                         float *corr, int32_t ldcorr) {
  KH_CHECK_INIT();
  KH_REQUIRE(N > 0 && O > 0 && I > 0);
  KH_REQUIRE(e && in && W_in && W_out && lde >= O && ldin >= I && ldw_in >= I && ldw_out >= I);
  KH_REQUIRE(momentum == 0.f || corr != nullptr);
  KH_REQUIRE(corr == nullptr || ldcorr >= I);
  // corr(OxI) = g * e^T(OxN) * in(NxI) + mom*corr, the weight step in the same epilogue
  GemmParams p = base_params(O, I, N, e, lde, in, ldin, corr, ldcorr);
  p.alpha = gscale;
  p.mom = momentum;
  p.lr = lr;
  p.l2 = l2;
  p.W = W_in;
  p.ldw = ldw_in;
  p.Wout = W_out;
  p.ldwout = ldw_out;
  return run_gemm<EPI_UPDATE>(s, false, false, p, /*allow_split=*/false);
}

KH_EXPORT int hipF_affine_update_bias(khStream_t s, int32_t N, int32_t O, const float *e, int32_t lde, float gscale,
                                      float momentum, float lr, float *bias, float *bcorr) {
  KH_CHECK_INIT();
  KH_REQUIRE(N > 0 && O > 0 && e && bias && lde >= O);
  KH_REQUIRE(momentum == 0.f || bcorr != nullptr);
  hipLaunchKernelGGL(bias_update_kernel, dim3((O + 63) / 64), dim3(1024), 0, kh::pick(s), e, lde, N, O, gscale,
                     momentum, lr, bias, bcorr);
  return kh::launch_status();
}

KH_EXPORT int hipF_affine_update(khStream_t s, int32_t N, int32_t O, int32_t I, const float *e, int32_t lde,
                                 const float *in, int32_t ldin, float gscale, float momentum, float lr, float l2,
                                 float *W, int32_t ldw, float *corr, int32_t ldcorr, float *bias, float *bcorr) {
  KH_REQUIRE(momentum == 0.f || (corr != nullptr && bcorr != nullptr));
  int rc = hipF_affine_update_weights(s, N, O, I, e, lde, in, ldin, gscale, momentum, lr, l2, W, ldw, W, ldw, corr,
                                      ldcorr);
  if (rc != KH_OK) return rc;
  return hipF_affine_update_bias(s, N, O, e, lde, gscale, momentum, lr, bias, bcorr);
}

KH_EXPORT int hipF_rbm_update_weights(khStream_t s, int32_t N, int32_t H, int32_t V, const float *pos_vis, int32_t ldpv,
                                      const float *pos_hid, int32_t ldph, const float *neg_vis, int32_t ldnv,
                                      const float *neg_hid, int32_t ldnh, float lr, float momentum, float l2, float *W,
                                      int32_t ldw, float *dW, int32_t lddw) {
  KH_CHECK_INIT();
  KH_REQUIRE(N > 0 && H > 0 && V > 0);
  KH_REQUIRE(pos_vis && pos_hid && neg_vis && neg_hid && W && dW);
  KH_REQUIRE(ldpv >= V && ldnv >= V && ldph >= H && ldnh >= H && ldw >= V && lddw >= V);
  const float g = lr / (float)N;
  // dW = -g * neg_hid^T neg_vis + mom * dW
  GemmParams p1 = base_params(H, V, N, neg_hid, ldnh, neg_vis, ldnv, dW, lddw);
  p1.alpha = -g;
  p1.beta = momentum;
  int rc = run_gemm<EPI_AXPBY>(s, false, false, p1, false);
  if (rc != KH_OK) return rc;
  // dW = +g * pos_hid^T pos_vis + dW - lr*l2*W ; W += dW
  GemmParams p2 = base_params(H, V, N, pos_hid, ldph, pos_vis, ldpv, dW, lddw);
  p2.alpha = g;
  p2.l2 = lr * l2;
  p2.W = W;
  p2.ldw = ldw;
  p2.Wout = W;
  p2.ldwout = ldw;
  return run_gemm<EPI_RBM>(s, false, false, p2, false);
}

KH_EXPORT int hipF_rbm_update_bias(khStream_t s, int32_t N, int32_t H, int32_t V, const float *pos_vis, int32_t ldpv,
                                   const float *pos_hid, int32_t ldph, const float *neg_vis, int32_t ldnv,
                                   const float *neg_hid, int32_t ldnh, float lr, float momentum, float *vis_bias,
                                   float *dvis_bias, float *hid_bias, float *dhid_bias) {
  KH_CHECK_INIT();
  KH_REQUIRE(N > 0 && H > 0 && V > 0);
  KH_REQUIRE(pos_vis && pos_hid && neg_vis && neg_hid && vis_bias && dvis_bias && hid_bias && dhid_bias);
  KH_REQUIRE(ldpv >= V && ldnv >= V && ldph >= H && ldnh >= H);
  const float g = lr / (float)N;
  hipStream_t st = kh::pick(s);
  hipLaunchKernelGGL(rbm_bias_kernel, dim3((V + 63) / 64), dim3(1024), 0, st, pos_vis, ldpv, neg_vis, ldnv, N, V, g,
                     momentum, vis_bias, dvis_bias);
  hipLaunchKernelGGL(rbm_bias_kernel, dim3((H + 63) / 64), dim3(1024), 0, st, pos_hid, ldph, neg_hid, ldnh, N, H, g,
                     momentum, hid_bias, dhid_bias);
  return kh::launch_status();
}

KH_EXPORT int hipF_rbm_update(khStream_t s, int32_t N, int32_t H, int32_t V, const float *pos_vis, int32_t ldpv,
                              const float *pos_hid, int32_t ldph, const float *neg_vis, int32_t ldnv,
                              const float *neg_hid, int32_t ldnh, float lr, float momentum, float l2, float *W,
                              int32_t ldw, float *dW, int32_t lddw, float *vis_bias, float *dvis_bias,
                              float *hid_bias, float *dhid_bias) {
  int rc = hipF_rbm_update_weights(s, N, H, V, pos_vis, ldpv, pos_hid, ldph, neg_vis, ldnv, neg_hid, ldnh, lr, momentum, l2,
                                   W, ldw, dW, lddw);
  if (rc != KH_OK) return rc;
  return hipF_rbm_update_bias(s, N, H, V, pos_vis, ldpv, pos_hid, ldph, neg_vis, ldnv, neg_hid, ldnh, lr, momentum, vis_bias,
                              dvis_bias, hid_bias, dhid_bias);
}
