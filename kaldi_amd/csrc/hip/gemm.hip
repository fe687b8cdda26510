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
#include "gemm_kernel.h"

namespace kh {
int softmax_slabs_launch(hipStream_t st, float *y, int ldy, const float *x, int ldx, int nslab, size_t slab_stride,
                         const float *bias, int rows, int cols, const XentArgs *xent);
}

namespace {
using namespace khg;

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
// hipK_gemm_row_invariant: only plans that sum a row's K range in one piece (no split over workgroups or waves), so
// that a row's result does not depend on how many other rows share the call
bool row_invariant = false;
bool env_read = false;

Plan make_plan(int M, int N, int K, int num_cu, bool allow_split, bool always_slabs, bool allow_wk, bool kmajor_whole_tile) {
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
  const double cyc_per_us = 2400.0;  // shader clock under sustained fp32 MFMA load (2.38-2.40 GHz by s_memtime / s_memrealtime)
  for (int c = 0; c < ncfg; ++c) {
    const TileCfg &t = kCfgs[c];
    if (t.wk > 1 && (!allow_wk || row_invariant)) continue;
    // (row-invariant plans, i.e. batched forward passes with thousands of rows: the 64x128 tile competes too)
    if (force_bm > 0 ? (t.bm != force_bm || t.bn != force_bn) : !(c == 0 || (t.wk > 1 && wk_on) || (row_invariant && c == 2)))
      continue;
    const int bkt = kBK * t.wk;
    const int ktiles = (K + bkt - 1) / bkt;
    const int tm = (M + t.bm - 1) / t.bm, tn = (N + t.bn - 1) / t.bn;
    for (int s = 1; s <= 16; s *= 2) {
      if (s > 1 && (!allow_split || row_invariant)) break;
      if (force_s > 0 && s != force_s) continue;
      if (s > ktiles) break;
      const int kt_per = (ktiles + s - 1) / s;
      const int s_eff = (ktiles + kt_per - 1) / kt_per;
      if (s_eff != s) continue;
      const double blocks = (double)tm * tn * s;
      const double kper = (double)kt_per * bkt;
      const double mfma = (double)t.bm * t.bn * kper * 2.0 / 256.0 * 1.08;
      // (K split over waves: same MFMA count per byte staged as the 64x64 tile is charged below)
      // Both operands K-contiguous and whole-tile fragments (forward products): the 64x64 K loop measures 1090-1110
      // cycles per 1024 of MFMA (in-kernel stamps), i.e. the MFMA term itself; the staging term below dates from
      // the group-by-group loop and still describes the other layouts.  (Output layer of the Switchboard net,
      // 256 x 9000 x 2048 + row kernel: 64x64 split 2 measures 92 us, the 32x64 split 2 the old term preferred 100.)
      const double mem = t.wk > 1 ? mfma * (1.28 / 1.08) : kmajor_whole_tile ? 0.0 : (double)(t.bm + t.bn) * kper * 4.0 / 12.5;
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
      if (row_invariant && kmajor_whole_tile && force_bm <= 0) {
        // Many rows, K unsplit: whole rounds of co-resident workgroups -- three per CU for the 64x64 kernel with
        // whole-tile fragments (152 registers), two for 64x128 -- each round as long as its workgroups' MFMAs
        // together.  (Sweep at 4096 rows: 2048-wide layer 64x64 270 us, 64x128 250; 9000-wide 1231 / 1103.)
        const double co_ri = c == 0 ? 3.0 : 2.0;
        cost = (ceil(blocks / (co_ri * num_cu)) * co_ri * mfma + t.fixed) / cyc_per_us + blocks * 1e-4;
      }
      if (s > 1 || always_slabs) cost += (s + 1.0) * (double)M * N * 4.0 / 3.5e6 + 2.5;
      // every doubling of the split doubles the workgroups that pay a prologue and an epilogue, which the fixed
      // term above undercounts for short K ranges (sweep of the 256 x 9000 x 2048 product: 64x64 split 4 measures
      // 5.7 us more than split 2 where the terms above put it 0.2 us ahead)
      for (int d = s; d > 1; d >>= 1) cost += 1.0;
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

// ---- optional per-launch timing -------------------------------------------------------------------
struct GemmTiming {
  bool enabled = false;
  std::vector<hipEvent_t> pool;                          // recycled events
  std::vector<std::pair<hipEvent_t, hipEvent_t> > spans;  // one per launch since the last read
  std::vector<double> span_flops;                          // algorithmic FLOPs of each
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
// The pair of events a timed launch carries (see KHG_LAUNCH in gemm_kernel.h): set for the one kernel launch inside a
// TimedLaunch scope, null otherwise.
}  // namespace
namespace khg {
hipEvent_t g_bind_start = nullptr, g_bind_stop = nullptr;
}
namespace {
struct TimedLaunch {
  hipEvent_t e0 = nullptr, e1 = nullptr;
  double fl_;
  TimedLaunch(hipStream_t, double fl) : fl_(fl) {
    if (!timing().enabled) return;
    e0 = timing_event();
    e1 = timing_event();
    if (e0 && e1) {
      khg::g_bind_start = e0;
      khg::g_bind_stop = e1;
      timing().flops += fl;
    }
  }
  ~TimedLaunch() {
    khg::g_bind_start = khg::g_bind_stop = nullptr;
    if (e0 && e1) {
      timing().spans.push_back(std::make_pair(e0, e1));
      timing().span_flops.push_back(fl_);
    }
  }
};

// tile grid, K split and operand footprints of `p` under `plan`
void apply_plan(GemmParams &p, const Plan &plan, bool a_kmajor, bool b_kmajor) {
  const int bm = kCfgs[plan.cfg].bm, bn = kCfgs[plan.cfg].bn;
  p.tiles_m = (p.M + bm - 1) / bm;
  p.tiles_n = (p.N + bn - 1) / bn;
  p.splitk = plan.splitk;
  p.k_per_split = plan.k_per_split;
  // rows x leading dimension of each operand as stored; buffer addressing needs < 2 GiB
  const double ab = ((double)((a_kmajor ? p.M : p.K) - 1) * p.lda + (a_kmajor ? p.K : p.M)) * 4.0;
  const double bb = ((double)((b_kmajor ? p.N : p.K) - 1) * p.ldb + (b_kmajor ? p.K : p.N)) * 4.0;
  const bool ok = ab < 2147483648.0 && bb < 2147483648.0 && p.K > 0;
  p.a_bytes = ok ? (int)ab : 0;
  p.b_bytes = ok ? (int)bb : 0;
}

// Runs op(A)*op(B) and the epilogue EPI.  `softmax` selects the row-softmax finish (always via
// slabs, since a row spans many tiles).
template <int EPI>
int run_gemm(khStream_t s, bool a_kmajor, bool b_kmajor, GemmParams p, bool allow_split, bool softmax = false,
             const kh::XentArgs *xent = nullptr) {
  hipStream_t st = kh::pick(s);
  const bool vec = kh::aligned16(p.A) && kh::aligned16(p.B) && (p.lda % 4 == 0) && (p.ldb % 4 == 0);
  Plan plan = make_plan(p.M, p.N, p.K, kh::ctx().num_cu > 0 ? kh::ctx().num_cu : 256, allow_split || softmax,
                        softmax, EPI != EPI_UPDATE && EPI != EPI_RBM,
                        a_kmajor && b_kmajor && EPI != EPI_UPDATE && EPI != EPI_RBM && KH_SCHED >= 2);
  apply_plan(p, plan, a_kmajor, b_kmajor);
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

// The update epilogue's 16-byte form needs W, Wout (and the momentum buffer) 16-byte aligned with leading
// dimensions and N multiples of 4, each below 2 GiB (buffer addressing); otherwise it stays element-wise.
void set_update_footprints(GemmParams &p) {
  static const bool off = getenv("KALDI_HIP_UPDATE_SCALAR_EPILOGUE") != nullptr;  // for A/B measurements
  auto bytes = [&](int ld) { return ((double)(p.M - 1) * ld + p.N) * 4.0; };
  const bool ok = !off && p.N % 4 == 0 && p.ldw % 4 == 0 && p.ldwout % 4 == 0 && kh::aligned16(p.W) && kh::aligned16(p.Wout) &&
                  (!p.C || (p.ldc % 4 == 0 && kh::aligned16(p.C))) && bytes(p.ldw) < 2147483648.0 &&
                  bytes(p.ldwout) < 2147483648.0 && (!p.C || bytes(p.ldc) < 2147483648.0);
  p.w_bytes = ok ? (int)bytes(p.ldw) : 0;
  p.wout_bytes = ok ? (int)bytes(p.ldwout) : 0;
  p.c_bytes = (ok && p.C) ? (int)bytes(p.ldc) : 0;
}

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
    if (mom != 0.f) bc = kh::mul_add(mom, bcorr[c], bc);
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
    float d = kh::mul_add(mom, dbias[c], -g * tn);
    d = kh::mul_add(g, tp, d);
    dbias[c] = d;
    bias[c] = d + bias[c];
  }
}

// Both bias statistics of the RBM step in one launch, 16-byte loads: workgroups [0, nv) take 64 visible units each,
// the rest 64 hidden units each.  Thread (g = tid / 16, quad = tid % 16) sums rows g, g + 16, ... of 4 adjacent
// columns of the positive and of the negative matrix -- 2 x 16 loads in flight per 256 rows, added in row order --
// and the 16 partial sums are combined in the order 0..15: the sums of rbm_bias_kernel, bit for bit.
// Needs 16-byte aligned rows (leading dimensions multiples of 4).
struct RbmBiasArgs {
  const float *pos, *neg;
  int ldp, ldn, D;
  float *bias, *dbias;
};
__global__ __launch_bounds__(256) void rbm_bias_vec_kernel(RbmBiasArgs vis, RbmBiasArgs hid, int nv, int N, float g, float mom) {
  typedef float v4f __attribute__((ext_vector_type(4)));
  __shared__ v4f pp[16][16];
  __shared__ v4f pn[16][16];
  const bool is_vis = (int)blockIdx.x < nv;
  const RbmBiasArgs &a = is_vis ? vis : hid;
  const int tile = is_vis ? blockIdx.x : blockIdx.x - nv;
  const int grp = threadIdx.x >> 4, quad = threadIdx.x & 15;
  const int c0 = tile * 64 + quad * 4;
  v4f ap = {0.f, 0.f, 0.f, 0.f}, an = {0.f, 0.f, 0.f, 0.f};
  if (c0 < a.D) {  // (D is a multiple of 4 here: whole quads)
    for (int r0 = grp; r0 < N; r0 += 256) {
      v4f xp[16], xn[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int r = r0 + 16 * u;
        const v4f z = {0.f, 0.f, 0.f, 0.f};
        xp[u] = r < N ? *reinterpret_cast<const v4f *>(a.pos + (size_t)r * a.ldp + c0) : z;
        xn[u] = r < N ? *reinterpret_cast<const v4f *>(a.neg + (size_t)r * a.ldn + c0) : z;
      }
#pragma unroll
      for (int u = 0; u < 16; ++u)
        if (r0 + 16 * u < N) {
          ap += xp[u];
          an += xn[u];
        }
    }
  }
  pp[grp][quad] = ap;
  pn[grp][quad] = an;
  __syncthreads();
  const int c = tile * 64 + threadIdx.x;
  if (threadIdx.x < 64 && c < a.D) {
    const float *fp = reinterpret_cast<const float *>(&pp[0][0]), *fn = reinterpret_cast<const float *>(&pn[0][0]);
    float tp = 0.f, tn = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      tp += fp[i * 64 + threadIdx.x];
      tn += fn[i * 64 + threadIdx.x];
    }
    float d = kh::mul_add(mom, a.dbias[c], -g * tn);
    d = kh::mul_add(g, tp, d);
    a.dbias[c] = d;
    a.bias[c] = d + a.bias[c];
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
      if (mom != 0.f) cc = kh::mul_add(mom, cv[i], cc);
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

// the same step on a range of a parameter arena that holds several layers: the hyper-parameters of an element are
// those of the segment that contains it (kernel arguments; a wave's elements lie in at most two segments)
struct SgdSegments {
  int n;
  unsigned begin[KH_MAX_SGD_SEGMENTS], end[KH_MAX_SGD_SEGMENTS];  // in units of 4 floats
  float g[KH_MAX_SGD_SEGMENTS], mom[KH_MAX_SGD_SEGMENTS], lr[KH_MAX_SGD_SEGMENTS], l2[KH_MAX_SGD_SEGMENTS];
};
__global__ __launch_bounds__(256) void sgd_apply_segments_kernel(SgdSegments sg, unsigned first4, unsigned count4,
                                                                 const float *grad, float *w, float *corr) {
  const unsigned stride = gridDim.x * 256;
  for (unsigned q = blockIdx.x * 256 + threadIdx.x; q < count4; q += stride) {
    const unsigned u = first4 + q;
    int k = -1;
    for (int i = 0; i < sg.n; ++i)
      if (u >= sg.begin[i] && u < sg.end[i]) k = i;
    if (k < 0) continue;
    const float g = sg.g[k], mom = sg.mom[k], lr = sg.lr[k], l2 = sg.l2[k];
    float gv[4], cv[4], wv[4];
    *reinterpret_cast<float4 *>(gv) = reinterpret_cast<const float4 *>(grad)[u];
    *reinterpret_cast<float4 *>(wv) = reinterpret_cast<const float4 *>(w)[u];
    if (mom != 0.f) *reinterpret_cast<float4 *>(cv) = reinterpret_cast<const float4 *>(corr)[u];
#pragma unroll
    for (int i = 0; i < 4; ++i) {  // (the arithmetic of affine_apply_kernel)
      float cc = g * gv[i];
      if (mom != 0.f) cc = kh::mul_add(mom, cv[i], cc);
      cv[i] = cc;
      float x = wv[i];
      if (l2 != 0.f) x = -l2 * x + x;
      wv[i] = -lr * cc + x;
    }
    reinterpret_cast<float4 *>(w)[u] = *reinterpret_cast<float4 *>(wv);
    if (corr) reinterpret_cast<float4 *>(corr)[u] = *reinterpret_cast<float4 *>(cv);
  }
}

__global__ void bias_apply_kernel(int O, const float *bgrad, float g, float mom, float lr, float *bias, float *bcorr) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= O) return;
  float bc = g * bgrad[c];
  if (mom != 0.f) bc = kh::mul_add(mom, bcorr[c], bc);
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

KH_EXPORT int hipF_sgd_apply_flat(khStream_t s, size_t count, const float *grad, float gscale, float momentum, float lr,
                                  float l2, float *w, float *corr) {
  KH_CHECK_INIT();
  if (count == 0) return KH_OK;
  KH_REQUIRE(grad && w && count < ((size_t)1 << 31));
  KH_REQUIRE(momentum == 0.f || corr != nullptr);
  const int n = (int)count;
  const bool vec = (n % 4 == 0) && kh::aligned16(grad) && kh::aligned16(w) && (!corr || kh::aligned16(corr));
  const int per = vec ? 4 : 1;
  const int gx = (n + 256 * per - 1) / (256 * per);
  hipStream_t st = kh::pick(s);
  // one long row: the element-wise formulas do not look at the position
  if (vec)
    hipLaunchKernelGGL((affine_apply_kernel<4>), dim3(gx, 1), dim3(256), 0, st, 1, n, grad, n, gscale, momentum, lr, l2, w, n,
                       corr, n);
  else
    hipLaunchKernelGGL((affine_apply_kernel<1>), dim3(gx, 1), dim3(256), 0, st, 1, n, grad, n, gscale, momentum, lr, l2, w, n,
                       corr, n);
  return kh::launch_status();
}

KH_EXPORT int hipF_sgd_apply_segments(khStream_t s, int32_t nseg, const khSgdSegment *segs, size_t first, size_t count,
                                      const float *grad, float *w, float *corr) {
  KH_CHECK_INIT();
  KH_REQUIRE(nseg >= 0 && nseg <= KH_MAX_SGD_SEGMENTS && (nseg == 0 || segs != nullptr));
  if (count == 0 || nseg == 0) return KH_OK;
  KH_REQUIRE(grad && w && kh::aligned16(grad) && kh::aligned16(w) && (!corr || kh::aligned16(corr)));
  KH_REQUIRE(first % 4 == 0 && count % 4 == 0 && (first + count) / 4 < ((size_t)1 << 32));
  SgdSegments sg;
  sg.n = nseg;
  for (int i = 0; i < nseg; ++i) {
    KH_REQUIRE(segs[i].begin % 4 == 0 && segs[i].end % 4 == 0 && segs[i].begin <= segs[i].end && segs[i].end / 4 < ((size_t)1 << 32));
    KH_REQUIRE(i == 0 || segs[i].begin >= segs[i - 1].end);
    KH_REQUIRE(segs[i].momentum == 0.f || corr != nullptr);
    sg.begin[i] = (unsigned)(segs[i].begin / 4);
    sg.end[i] = (unsigned)(segs[i].end / 4);
    sg.g[i] = segs[i].gscale;
    sg.mom[i] = segs[i].momentum;
    sg.lr[i] = segs[i].lr;
    sg.l2[i] = segs[i].l2;
  }
  const unsigned count4 = (unsigned)(count / 4);
  unsigned gx = (count4 + 255) / 256;
  if (gx > 4096) gx = 4096;  // a few float4 per thread at most for the large shards: enough loads in flight per CU
  hipLaunchKernelGGL(sgd_apply_segments_kernel, dim3(gx), dim3(256), 0, kh::pick(s), sg, (unsigned)(first / 4), count4, grad, w,
                     corr);
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

KH_EXPORT int hipK_gemm_row_invariant(int on) {
  row_invariant = on != 0;
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
    if (i > 0) {  // (begin of launch i = its end, measured from the first launch's begin, less its length)
      KH_HIP(hipEventElapsedTime(&off, t.spans[0].first, t.spans[i].second));
      off -= one;
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
  t.span_flops.clear();
  t.flops = 0.0;
  return KH_OK;
}

KH_EXPORT int hipK_gemm_timing_spans(double *start_ms, double *dur_ms, double *gflop, int cap, int *count) {
  GemmTiming &t = timing();
  KH_REQUIRE(start_ms && dur_ms && count && cap >= 0);
  const int n = (int)t.spans.size() < cap ? (int)t.spans.size() : cap;
  for (int i = 0; i < n; ++i) {
    KH_HIP(hipEventSynchronize(t.spans[i].second));
    float one = 0.f, off = 0.f;
    KH_HIP(hipEventElapsedTime(&one, t.spans[i].first, t.spans[i].second));
    if (i > 0) {
      KH_HIP(hipEventElapsedTime(&off, t.spans[0].first, t.spans[i].second));
      off -= one;
    }
    start_ms[i] = off;
    dur_ms[i] = one;
    if (gflop) gflop[i] = i < (int)t.span_flops.size() ? t.span_flops[i] * 1e-9 : 0.0;
  }
  *count = n;
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

KH_EXPORT int hipF_affine_forward_sample(khStream_t s, int32_t N, int32_t O, int32_t I, const float *in, int32_t ldin,
                                         const float *W, int32_t ldw, const float *bias, float *probs, int32_t ldprobs,
                                         float *states, int32_t ldstates, uint32_t *z1, uint32_t *z2, uint32_t *z3,
                                         uint32_t *z4) {
  KH_CHECK_INIT();
  KH_REQUIRE(N >= 0 && O > 0 && I > 0);
  if (N == 0) return KH_OK;
  KH_REQUIRE(in && W && bias && probs && states && z1 && z2 && z3 && z4 && ldin >= I && ldw >= I && ldprobs >= O && ldstates == ldprobs);
  GemmParams p = base_params(N, O, I, in, ldin, W, ldw, probs, ldprobs);
  p.bias = bias;
  p.act = KH_ACT_SIGMOID;
  p.C2 = states;
  p.z1 = z1; p.z2 = z2; p.z3 = z3; p.z4 = z4;
  return run_gemm<EPI_BIAS_SAMPLE>(s, true, true, p, /*allow_split=*/false);
}

namespace {
struct ChainState {
  unsigned *counters = nullptr;  // device: kMaxChain x 64 arrival counters, only ever growing
  int *fail = nullptr;           // host-mapped: set by a workgroup whose wait ran out
  unsigned epoch = 0;
};
ChainState &chain_state() {
  static ChainState c;
  return c;
}
}  // namespace

KH_EXPORT int hipF_affine_forward_chain(khStream_t s, int act, int32_t N, int32_t count, const khAffineForward *layers) {
  KH_CHECK_INIT();
  KH_REQUIRE(act == KH_ACT_NONE || act == KH_ACT_SIGMOID);
  KH_REQUIRE(N >= 0 && count >= 0 && (count == 0 || layers != nullptr));
  for (int i = 0; i < count; ++i) {
    const khAffineForward &l = layers[i];
    KH_REQUIRE(l.O > 0 && l.I > 0 && l.in && l.W && l.bias && l.out && l.ldin >= l.I && l.ldout >= l.O && l.ldw >= l.I);
    KH_REQUIRE(i == 0 || (l.in == layers[i - 1].out && l.ldin == layers[i - 1].ldout && l.I == layers[i - 1].O));
  }
  if (N == 0 || count == 0) return KH_OK;
  static const bool off = getenv("KALDI_HIP_NO_CHAIN") != nullptr;  // for A/B measurements
  hipStream_t st = kh::pick(s);
  const int cus = kh::ctx().num_cu > 0 ? kh::ctx().num_cu : 256;
  // the chained launch, where every layer qualifies
  GemmChain g;
  memset(&g, 0, sizeof(g));
  bool ok = !off && count >= 2 && count <= kMaxChain && force_bm <= 0 && !row_invariant;
  int cfg = -1;
  double flops = 0.0;
  for (int i = 0; ok && i < count; ++i) {
    const khAffineForward &l = layers[i];
    GemmParams p = base_params(N, l.O, l.I, l.in, l.ldin, l.W, l.ldw, l.out, l.ldout);
    p.bias = l.bias;
    p.act = act;
    const bool vec = kh::aligned16(p.A) && kh::aligned16(p.B) && (p.lda % 4 == 0) && (p.ldb % 4 == 0);
    Plan plan = make_plan(p.M, p.N, p.K, cus, true, false, true, KH_SCHED >= 2);
    apply_plan(p, plan, true, true);
    const int wk = kCfgs[plan.cfg].wk;
    ok = vec && p.a_bytes > 0 && wk > 1 && plan.splitk == 1 && p.K % (kBK * wk) == 0 && p.tiles_m * p.tiles_n <= cus &&
         p.tiles_m <= 64 && (cfg < 0 || (plan.cfg == cfg && p.tiles_m == g.p[0].tiles_m && p.tiles_n == g.p[0].tiles_n));
    cfg = plan.cfg;
    p.n_fastest = 0;
    g.p[i] = p;
    flops += 2.0 * p.M * p.N * p.K;
  }
  if (!ok) {
    for (int i = 0; i < count; ++i) {
      const khAffineForward &l = layers[i];
      const int rc = hipF_affine_forward(s, act, KH_TRANS, N, l.O, l.I, l.in, l.ldin, l.W, l.ldw, l.bias, l.out, l.ldout);
      if (rc != KH_OK) return rc;
    }
    return KH_OK;
  }
  ChainState &cs = chain_state();
  if (!cs.counters) {
    KH_HIP(hipMalloc((void **)&cs.counters, kMaxChain * 64 * sizeof(unsigned)));
    KH_HIP(hipMemsetAsync(cs.counters, 0, kMaxChain * 64 * sizeof(unsigned), st));
    KH_HIP(hipHostMalloc((void **)&cs.fail, sizeof(int), hipHostMallocMapped));
    *cs.fail = 0;
  }
  if (*cs.fail != 0) {
    snprintf(kh::ctx().last_msg, sizeof(kh::ctx().last_msg), "hipF_affine_forward_chain: a hand-off wait ran out in layer %d of an earlier launch",
             *cs.fail);
    return KH_ERR_INVALID_ARG;
  }
  if (cs.epoch >= (1u << 24)) {  // counters start over long before they could wrap
    KH_HIP(hipMemsetAsync(cs.counters, 0, kMaxChain * 64 * sizeof(unsigned), st));
    cs.epoch = 0;
  }
  g.n = count;
  g.epoch = ++cs.epoch;
  g.counters = cs.counters;
  g.fail = cs.fail;
  {
    TimedLaunch tl(st, flops);
    launch_forward_chain(st, g, cfg);
  }
  return kh::launch_status();
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
  set_update_footprints(p);
  return run_gemm<EPI_UPDATE>(s, false, false, p, /*allow_split=*/false);
}

KH_EXPORT int hipF_affine_update_fused(khStream_t s, int32_t N, int32_t O, int32_t I, const float *e, int32_t lde,
                                       const float *in, int32_t ldin, float gscale, float momentum, float lr, float l2,
                                       const float *W_in, int32_t ldw_in, float *W_out, int32_t ldw_out, float *corr,
                                       int32_t ldcorr, float *bias, float *bcorr) {
  KH_CHECK_INIT();
  KH_REQUIRE(N > 0 && O > 0 && I > 0);
  KH_REQUIRE(e && in && W_in && W_out && bias && lde >= O && ldin >= I && ldw_in >= I && ldw_out >= I);
  KH_REQUIRE(momentum == 0.f || (corr != nullptr && bcorr != nullptr));
  KH_REQUIRE(corr == nullptr || ldcorr >= I);
  GemmParams p = base_params(O, I, N, e, lde, in, ldin, corr, ldcorr);
  p.alpha = gscale;
  p.mom = momentum;
  p.lr = lr;
  p.l2 = l2;
  p.W = W_in;
  p.ldw = ldw_in;
  p.Wout = W_out;
  p.ldwout = ldw_out;
  set_update_footprints(p);
  p.bias_io = bias;
  p.bcorr = bcorr;
  p.bias_tiles = (O + 63) / 64;  // ride in the GEMM's launch
  return run_gemm<EPI_UPDATE>(s, false, false, p, /*allow_split=*/false);
}

bool group_big = getenv("KALDI_HIP_GROUP_TILE64") == nullptr;  // (the variable: for A/B measurements)

KH_EXPORT int hipF_affine_update_group(khStream_t s, int32_t count, const khAffineUpdate *items) {
  KH_CHECK_INIT();
  KH_REQUIRE(count >= 0 && (count == 0 || items != nullptr));
  static const bool off = getenv("KALDI_HIP_NO_UPDATE_GROUP") != nullptr;  // for A/B measurements
  bool big = group_big;
  hipStream_t st = kh::pick(s);
  GemmGroup g;
  memset(&g, 0, sizeof(g));
  double flops = 0.0;
  auto flush = [&]() -> int {
    if (g.n == 0) return KH_OK;
    {
      TimedLaunch tl(st, flops);
      launch_update_group(st, g, big);
    }
    g.n = 0;
    g.first[0] = 0;
    flops = 0.0;
    return kh::launch_status();
  };
  // every item is checked before anything is launched: a refused call has changed nothing
  for (int i = 0; i < count; ++i) {
    const khAffineUpdate &u = items[i];
    KH_REQUIRE(u.N > 0 && u.O > 0 && u.I > 0);
    KH_REQUIRE(u.e && u.in && u.W_in && u.W_out && u.bias && u.lde >= u.O && u.ldin >= u.I && u.ldw_in >= u.I && u.ldw_out >= u.I);
    KH_REQUIRE(u.momentum == 0.f || (u.corr != nullptr && u.bcorr != nullptr));
    KH_REQUIRE(u.corr == nullptr || u.ldcorr >= u.I);
  }
  // 128 x 128 tiles where the call fills at least two rounds of their 2-per-CU slots; a small network's updates
  // (RM-sized: 320 such tiles) would all start together and run one long workgroup lifetime: 64 x 64 tiles there
  // (RM step 0.1227 -> 0.1175 ms).
  if (big) {
    const int cus = kh::ctx().num_cu > 0 ? kh::ctx().num_cu : 256;
    long long big_tiles = 0;
    for (int i = 0; i < count; ++i) big_tiles += (long long)((items[i].O + 127) / 128) * ((items[i].I + 127) / 128);
    if (big_tiles < 4LL * cus) big = false;
  }
  for (int i = 0; i < count; ++i) {
    const khAffineUpdate &u = items[i];
    GemmParams p = base_params(u.O, u.I, u.N, u.e, u.lde, u.in, u.ldin, u.corr, u.ldcorr);
    p.alpha = u.gscale;
    p.mom = u.momentum;
    p.lr = u.lr;
    p.l2 = u.l2;
    p.W = u.W_in;
    p.ldw = u.ldw_in;
    p.Wout = u.W_out;
    p.ldwout = u.ldw_out;
    set_update_footprints(p);
    p.bias_io = u.bias;
    p.bcorr = u.bcorr;
    p.bias_tiles = (u.O + 63) / 64;
    const bool vec = kh::aligned16(p.A) && kh::aligned16(p.B) && (p.lda % 4 == 0) && (p.ldb % 4 == 0);
    Plan plan = {big ? 3 : 0, 1, ((p.K + kBK - 1) / kBK) * kBK};  // 128 x 128 or 64 x 64 tiles, K whole
    apply_plan(p, plan, false, false);
    // an XCD's run of tiles: leave the operand with more tiles sliced (the 9000-output layer: the error matrix)
    p.n_fastest = p.tiles_m > 2 * p.tiles_n ? 1 : 0;
    if (off || !vec || p.a_bytes == 0 || force_bm > 0) {
      // not for the grouped kernel: on its own, in order
      int rc = flush();
      if (rc != KH_OK) return rc;
      rc = run_gemm<EPI_UPDATE>(s, false, false, p, /*allow_split=*/false);
      if (rc != KH_OK) return rc;
      continue;
    }
    g.p[g.n] = p;
    const int blocks = p.tiles_m * p.tiles_n;
    g.first[g.n + 1] = g.first[g.n] + ((blocks + 7) & ~7);
    flops += 2.0 * p.M * p.N * p.K;
    if (++g.n == kMaxGroup) {
      const int rc = flush();
      if (rc != KH_OK) return rc;
    }
  }
  return flush();
}

KH_EXPORT int hipF_affine_gradient_group(khStream_t s, int32_t count, const khAffineGradient *items) {
  return hipF_affine_gradient_group_marked(s, count, items, nullptr);
}

KH_EXPORT int hipF_affine_gradient_group_marked(khStream_t s, int32_t count, const khAffineGradient *items, khEvent_t done) {
  KH_CHECK_INIT();
  KH_REQUIRE(count >= 0 && (count == 0 || items != nullptr));
  hipStream_t st = kh::pick(s);
  for (int i = 0; i < count; ++i) {
    const khAffineGradient &u = items[i];
    KH_REQUIRE(u.N > 0 && u.O > 0 && u.I > 0);
    KH_REQUIRE(u.e && u.in && u.gW && u.gb && u.lde >= u.O && u.ldin >= u.I && u.ldg >= u.I);
  }
  // tiles as in hipF_affine_update_group: 128 x 128 where the call fills two rounds of their slots
  static const int big_rounds = getenv("KALDI_HIP_GRAD_GROUP_BIG_TILES") ? atoi(getenv("KALDI_HIP_GRAD_GROUP_BIG_TILES")) : 4;
  bool big = group_big;
  if (big) {
    const int cus = kh::ctx().num_cu > 0 ? kh::ctx().num_cu : 256;
    long long big_tiles = 0;
    for (int i = 0; i < count; ++i) big_tiles += (long long)((items[i].O + 127) / 128) * ((items[i].I + 127) / 128);
    if (big_tiles < (long long)big_rounds * cus) big = false;
  }
  GemmGroup g;
  memset(&g, 0, sizeof(g));
  double flops = 0.0;
  bool marked = false;  // `done` has been bound to a launch (only ever the call's last one)
  auto flush = [&](bool last) -> int {
    if (g.n == 0) return KH_OK;
    {
      TimedLaunch tl(st, flops);
      launch_grad_group(st, g, big, (last && !timing().enabled) ? (hipEvent_t)done : nullptr);
      marked = last && !timing().enabled && done != nullptr;
    }
    g.n = 0;
    g.first[0] = 0;
    flops = 0.0;
    return kh::launch_status();
  };
  for (int i = 0; i < count; ++i) {
    const khAffineGradient &u = items[i];
    GemmParams p = base_params(u.O, u.I, u.N, u.e, u.lde, u.in, u.ldin, u.gW, u.ldg);
    const double cbytes = ((double)(u.O - 1) * u.ldg + u.I) * 4.0;
    const bool cvec = u.I % 4 == 0 && u.ldg % 4 == 0 && kh::aligned16(u.gW) && cbytes < 2147483648.0;
    p.c_bytes = cvec ? (int)cbytes : 0;
    p.bias_io = u.gb;
    p.bias_tiles = (u.O + 63) / 64;
    const bool vec = kh::aligned16(p.A) && kh::aligned16(p.B) && (p.lda % 4 == 0) && (p.ldb % 4 == 0);
    Plan plan = {big ? 3 : 0, 1, ((p.K + kBK - 1) / kBK) * kBK};
    apply_plan(p, plan, false, false);
    p.n_fastest = p.tiles_m > 2 * p.tiles_n ? 1 : 0;
    if (!vec || !cvec || p.a_bytes == 0 || force_bm > 0) {
      int rc = flush(false);
      if (rc != KH_OK) return rc;
      rc = hipF_gemm(s, KH_TRANS, KH_NO_TRANS, u.O, u.I, u.N, 1.0f, u.e, u.lde, u.in, u.ldin, 0.0f, u.gW, u.ldg);
      if (rc != KH_OK) return rc;
      ::MatrixDim d = {u.N, u.O, u.lde};
      rc = hipF_add_row_sum_mat(s, 1.0f, u.e, d, 0.0f, u.gb);
      if (rc != KH_OK) return rc;
      continue;
    }
    g.p[g.n] = p;
    const int blocks = p.tiles_m * p.tiles_n;
    g.first[g.n + 1] = g.first[g.n] + ((blocks + 7) & ~7);
    flops += 2.0 * p.M * p.N * p.K;
    if (++g.n == kMaxGroup) {
      const int rc = flush(i == count - 1);
      if (rc != KH_OK) return rc;
    }
  }
  const int rc = flush(true);
  if (rc != KH_OK) return rc;
  // (nothing left for the last launch to carry the mark, or launches bracketed by timing events: a mark of its own)
  if (done != nullptr && !marked) KH_HIP(hipEventRecord((hipEvent_t)done, st));
  return KH_OK;
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
  return hipF_affine_update_fused(s, N, O, I, e, lde, in, ldin, gscale, momentum, lr, l2, W, ldw, W, ldw, corr, ldcorr, bias,
                                  bcorr);
}

namespace {
// Vectors of the bias half, for the launch that can carry it (rbm_update_weights sets `done` when it did).
struct RbmBiasVectors {
  float *vis_bias, *dvis_bias, *hid_bias, *dhid_bias;
  bool done;
  double *mse_loss;  // not null: the reconstruction error rides along too (mse_done)
  bool mse_done;
};
int rbm_update_weights(khStream_t s, int32_t N, int32_t H, int32_t V, const float *pos_vis, int32_t ldpv,
                       const float *pos_hid, int32_t ldph, const float *neg_vis, int32_t ldnv, const float *neg_hid,
                       int32_t ldnh, float lr, float momentum, float l2, float *W, int32_t ldw, float *dW, int32_t lddw,
                       RbmBiasVectors *bv) {
  KH_CHECK_INIT();
  KH_REQUIRE(N > 0 && H > 0 && V > 0);
  KH_REQUIRE(pos_vis && pos_hid && neg_vis && neg_hid && W && dW);
  KH_REQUIRE(ldpv >= V && ldnv >= V && ldph >= H && ldnh >= H && ldw >= V && lddw >= V);
  const float g = lr / (float)N;
  // Both statistics in one pass over W and dW where the operands allow it: the K loop runs over the rows of the
  // negative pair, then over those of the positive pair, each into accumulators of its own, and the epilogue
  // applies the two formulas below in order (GemmParams::nk1).  Needs whole 32-row K-tiles, 16-byte aligned
  // operands with equal leading dimensions per pair, and the 64 x 64 tile (whose group-by-group loop carries the
  // product switch).  Measured at 2048 x 2048, bunch 256: 61 us for the two launches, 41 us for the one.
  static const bool two_launches = getenv("KALDI_HIP_RBM_TWO_LAUNCHES") != nullptr;  // for A/B measurements
  if (!two_launches && N % kBK == 0 && ldnh == ldph && ldnv == ldpv && ldnh % 4 == 0 && ldnv % 4 == 0 && kh::aligned16(neg_hid) &&
      kh::aligned16(pos_hid) && kh::aligned16(neg_vis) && kh::aligned16(pos_vis) && force_bm <= 0) {
    GemmParams p = base_params(H, V, N, neg_hid, ldnh, neg_vis, ldnv, dW, lddw);
    Plan plan = {0, 1, N};
    apply_plan(p, plan, false, false);  // (tile grid and operand footprints: those of ONE product)
    if (p.a_bytes != 0) {
      p.A2 = pos_hid;
      p.B2 = pos_vis;
      p.alpha = -g;
      p.beta = momentum;
      p.alpha2 = g;
      p.l2 = lr * l2;
      p.W = W;
      p.ldw = ldw;
      p.Wout = W;
      p.ldwout = ldw;
      set_update_footprints(p);  // 16-byte epilogue where W and dW allow it, else element by element
      p.nk1 = N / kBK;
      p.K = 2 * N;
      p.k_per_split = 2 * N;
      if (bv && H % 4 == 0 && V % 4 == 0) {
        // both bias statistics by workgroups at the end of this launch (rbm_bias_tile): no launch of their own
        p.bias_io = bv->hid_bias;
        p.bcorr = bv->dhid_bias;
        p.bias2_io = bv->vis_bias;
        p.bcorr2 = bv->dvis_bias;
        p.bias_tiles = (H + 63) / 64 + (V + 63) / 64;
        bv->done = true;
      }
      hipStream_t st = kh::pick(s);
      if (bv && bv->done && bv->mse_loss) {
        // one more workgroup per row: the reconstruction error 1/2 * sum((neg_vis - pos_vis)^2), as hipF_mse_eval
        void *ws;
        int rc = kh::ensure_workspace(st, (size_t)N * 4, &ws);
        if (rc != KH_OK) return rc;
        rc = kh::stream_ticket(st, &p.mse_ticket);
        if (rc != KH_OK) return rc;
        p.mse_rows = N;
        p.mse_row_sq = (float *)ws;
        p.mse_loss = bv->mse_loss;
        bv->mse_done = true;
      }
      TimedLaunch tl(st, 4.0 * H * V * N);
      launch_layout<EPI_RBM>(st, p, 0, true, false, false);
      return kh::launch_status();
    }
  }
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
}  // namespace

KH_EXPORT int hipF_rbm_update_weights(khStream_t s, int32_t N, int32_t H, int32_t V, const float *pos_vis, int32_t ldpv,
                                      const float *pos_hid, int32_t ldph, const float *neg_vis, int32_t ldnv,
                                      const float *neg_hid, int32_t ldnh, float lr, float momentum, float l2, float *W,
                                      int32_t ldw, float *dW, int32_t lddw) {
  return rbm_update_weights(s, N, H, V, pos_vis, ldpv, pos_hid, ldph, neg_vis, ldnv, neg_hid, ldnh, lr, momentum, l2, W, ldw,
                            dW, lddw, nullptr);
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
  if (V % 4 == 0 && H % 4 == 0 && ldpv % 4 == 0 && ldnv % 4 == 0 && ldph % 4 == 0 && ldnh % 4 == 0 && kh::aligned16(pos_vis) &&
      kh::aligned16(neg_vis) && kh::aligned16(pos_hid) && kh::aligned16(neg_hid)) {
    const RbmBiasArgs av = {pos_vis, neg_vis, ldpv, ldnv, V, vis_bias, dvis_bias};
    const RbmBiasArgs ah = {pos_hid, neg_hid, ldph, ldnh, H, hid_bias, dhid_bias};
    const int nv = (V + 63) / 64;
    hipLaunchKernelGGL(rbm_bias_vec_kernel, dim3(nv + (H + 63) / 64), dim3(256), 0, st, av, ah, nv, N, g, momentum);
    return kh::launch_status();
  }
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
  KH_REQUIRE(vis_bias && dvis_bias && hid_bias && dhid_bias);
  static const bool own_launch = getenv("KALDI_HIP_RBM_BIAS_LAUNCH") != nullptr;  // for A/B measurements
  RbmBiasVectors bv = {vis_bias, dvis_bias, hid_bias, dhid_bias, false, nullptr, false};
  int rc = rbm_update_weights(s, N, H, V, pos_vis, ldpv, pos_hid, ldph, neg_vis, ldnv, neg_hid, ldnh, lr, momentum, l2, W, ldw,
                              dW, lddw, own_launch ? nullptr : &bv);
  if (rc != KH_OK || bv.done) return rc;
  return hipF_rbm_update_bias(s, N, H, V, pos_vis, ldpv, pos_hid, ldph, neg_vis, ldnv, neg_hid, ldnh, lr, momentum, vis_bias,
                              dvis_bias, hid_bias, dhid_bias);
}

KH_EXPORT int hipF_rbm_update_mse(khStream_t s, int32_t N, int32_t H, int32_t V, const float *pos_vis, int32_t ldpv,
                                  const float *pos_hid, int32_t ldph, const float *neg_vis, int32_t ldnv,
                                  const float *neg_hid, int32_t ldnh, float lr, float momentum, float l2, float *W,
                                  int32_t ldw, float *dW, int32_t lddw, float *vis_bias, float *dvis_bias,
                                  float *hid_bias, float *dhid_bias, double *mse_loss) {
  KH_REQUIRE(vis_bias && dvis_bias && hid_bias && dhid_bias && mse_loss);
  static const bool own_launch = getenv("KALDI_HIP_RBM_BIAS_LAUNCH") != nullptr;  // for A/B measurements
  RbmBiasVectors bv = {vis_bias, dvis_bias, hid_bias, dhid_bias, false, mse_loss, false};
  if (!own_launch) {
    int rc = rbm_update_weights(s, N, H, V, pos_vis, ldpv, pos_hid, ldph, neg_vis, ldnv, neg_hid, ldnh, lr, momentum, l2, W,
                                ldw, dW, lddw, &bv);
    if (rc != KH_OK || (bv.done && bv.mse_done)) return rc;
  }
  // the error in a launch of its own (before the weights where none of them has run yet; after is as good: it reads
  // the reconstruction and the data only)
  MatrixDim d = {N, V, ldnv};
  KH_REQUIRE(ldnv == ldpv);
  int rc = hipF_mse_eval(s, neg_vis, pos_vis, nullptr, nullptr, mse_loss, d);
  if (rc != KH_OK || bv.done) return rc;
  if (own_launch) {
    rc = hipF_rbm_update_weights(s, N, H, V, pos_vis, ldpv, pos_hid, ldph, neg_vis, ldnv, neg_hid, ldnh, lr, momentum, l2, W,
                                 ldw, dW, lddw);
    if (rc != KH_OK) return rc;
  }
  return hipF_rbm_update_bias(s, N, H, V, pos_vis, ldpv, pos_hid, ldph, neg_vis, ldnv, neg_hid, ldnh, lr, momentum, vis_bias,
                              dvis_bias, hid_bias, dhid_bias);
}
