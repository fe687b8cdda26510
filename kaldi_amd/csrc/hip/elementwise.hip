// Element-wise kernels of the CuMatrix / cu:: surface (src/cudamatrix/cu-kernels.cu:120-454,
// 509-596, 672-695).  They are HBM-bound, so every one of them is the same streaming loop:
// 16 bytes per lane where the layout allows it, one wave-instruction = 1 KiB, consecutive
// lanes on consecutive addresses, rows spread over the grid's y dimension.
#include "common.h"

namespace {

struct EwArgs {
  float *dst;        // written (and read when Op::kReadDst)
  const float *a;    // optional same-shaped input
  const float *b;    // optional second same-shaped input
  const float *colv; // optional vector indexed by column
  const float *rowv; // optional vector indexed by row
  float p0, p1;
};

// One functor per operation: which inputs it reads and the scalar formula.
#define KH_OP(NAME, READ_DST, USE_A, USE_B, USE_COLV, USE_ROWV, EXPR)                              \
  struct NAME {                                                                                    \
    static constexpr bool kReadDst = READ_DST, kUseA = USE_A, kUseB = USE_B, kColV = USE_COLV,     \
                          kRowV = USE_ROWV;                                                        \
    __device__ static __forceinline__ float apply(float m, float a, float b, float cv, float rv,   \
                                                  float p0, float p1) {                            \
      return EXPR;                                                                                 \
    }                                                                                              \
  }

KH_OP(OpSetConst, false, false, false, false, false, p0);
KH_OP(OpBinarize, true, false, false, false, false, (m > p0) ? 1.0f : 0.0f);
KH_OP(OpAddConst, true, false, false, false, false, m + p0);
KH_OP(OpPower, true, false, false, false, false, powf(m, p0));
KH_OP(OpRound, true, false, false, false, false, roundf(m));
KH_OP(OpScale, true, false, false, false, false, m *p0);
KH_OP(OpFloor, true, false, false, false, false, (m < p0) ? p0 : m);
// two sequential tests like the reference (low first, then high)
KH_OP(OpTruncate, true, false, false, false, false, ((m < p0 ? p0 : m) > p1) ? p1 : (m < p0 ? p0 : m));
KH_OP(OpLog, true, false, false, false, false, logf(m));
KH_OP(OpExp, true, false, false, false, false, expf(m));
KH_OP(OpInvert, true, false, false, false, false, 1.0f / m);
KH_OP(OpMulElements, true, true, false, false, false, m *a);
KH_OP(OpDivElements, true, true, false, false, false, m / a);
KH_OP(OpMulColsVec, true, false, false, true, false, m *cv);
KH_OP(OpMulRowsVec, true, false, false, false, true, m *rv);
KH_OP(OpDivColsVec, true, false, false, true, false, m *(1.0f / cv));
KH_OP(OpDivRowsVec, true, false, false, false, true, m *(1.0f / rv));
// beta == 0 must not propagate NaN/inf from uninitialised dst: handled by OpAddMatBeta0 & co.
KH_OP(OpAddMat, true, true, false, false, false, p0 *a + p1 * m);
KH_OP(OpAddMatBeta0, false, true, false, false, false, p0 *a);
KH_OP(OpAddVecToCols, true, false, false, false, true, p0 *rv + p1 * m);
KH_OP(OpAddVecToColsBeta0, false, false, false, false, true, p0 *rv);
KH_OP(OpAddVecToRows, true, false, false, true, false, p0 *cv + p1 * m);
KH_OP(OpAddVecToRowsBeta0, false, false, false, true, false, p0 *cv);
KH_OP(OpRelu, false, true, false, false, false, (a > 0.0f) ? a : 0.0f);
KH_OP(OpDiffRelu, false, true, true, false, false, (b > 0.0f) ? a : 0.0f);
KH_OP(OpSoftRelu, false, true, false, false, false, (a > 4.0f) ? a : logf(1.0f + expf(a)));
KH_OP(OpSigmoid, false, true, false, false, false, kh::sigmoidf(a));
KH_OP(OpDiffSigmoid, false, true, true, false, false, b *(1.0f - b) * a);
KH_OP(OpBinarizeProbs, false, true, true, false, false, (a > b) ? 1.0f : 0.0f);

struct OpDiffSoftRelu {
  static constexpr bool kReadDst = false, kUseA = true, kUseB = true, kColV = false, kRowV = false;
  __device__ static __forceinline__ float apply(float, float e, float x, float, float, float, float) {
    if (x > 4.0f) return e;
    float f = expf(x);
    return e * f / (1.0f + f);
  }
};

struct OpLogAddExp {  // double evaluation, cu-kernels.cu:362-380
  static constexpr bool kReadDst = true, kUseA = true, kUseB = false, kColV = false, kRowV = false;
  __device__ static __forceinline__ float apply(float m, float a, float, float, float, float, float) {
    double fa = a, fb = m, mx, sum;
    if (fa > fb) {
      sum = 1.0 + exp(fb - fa);
      mx = fa;
    } else {
      sum = exp(fa - fb) + 1.0;
      mx = fb;
    }
    return (float)(mx + log(sum));
  }
};

template <class Op, int VEC>
__global__ __launch_bounds__(256) void ew_kernel(EwArgs A, MatrixDim d) {
  const int c = (blockIdx.x * 256 + threadIdx.x) * VEC;
  if (c >= d.cols) return;
  float cv[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) cv[i] = Op::kColV ? A.colv[c + i] : 0.f;
  for (int r = blockIdx.y; r < d.rows; r += gridDim.y) {
    const size_t idx = (size_t)r * d.stride + c;
    const float rv = Op::kRowV ? A.rowv[r] : 0.f;
    float m[VEC], a[VEC], b[VEC];
    if (VEC == 4) {
      if (Op::kReadDst) *reinterpret_cast<float4 *>(m) = *reinterpret_cast<const float4 *>(A.dst + idx);
      if (Op::kUseA) *reinterpret_cast<float4 *>(a) = *reinterpret_cast<const float4 *>(A.a + idx);
      if (Op::kUseB) *reinterpret_cast<float4 *>(b) = *reinterpret_cast<const float4 *>(A.b + idx);
    } else {
      if (Op::kReadDst) m[0] = A.dst[idx];
      if (Op::kUseA) a[0] = A.a[idx];
      if (Op::kUseB) b[0] = A.b[idx];
    }
    float o[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i)
      o[i] = Op::apply(Op::kReadDst ? m[i] : 0.f, Op::kUseA ? a[i] : 0.f, Op::kUseB ? b[i] : 0.f, cv[i], rv,
                       A.p0, A.p1);
    if (VEC == 4)
      *reinterpret_cast<float4 *>(A.dst + idx) = *reinterpret_cast<float4 *>(o);
    else
      A.dst[idx] = o[0];
  }
}

// 16-byte accesses of data that is touched once (-DKH_EW_NT=0: plain accesses, for A/B builds)
typedef float ew_f4 __attribute__((ext_vector_type(4)));
#ifndef KH_EW_NT
#define KH_EW_NT 1
#endif
__device__ __forceinline__ float4 ld_stream(const float *p) {
#if KH_EW_NT
  const ew_f4 v = __builtin_nontemporal_load(reinterpret_cast<const ew_f4 *>(p));
  return make_float4(v.x, v.y, v.z, v.w);
#else
  return *reinterpret_cast<const float4 *>(p);
#endif
}
__device__ __forceinline__ void st_stream(float *p, float4 o) {
#if KH_EW_NT
  ew_f4 v = {o.x, o.y, o.z, o.w};
  __builtin_nontemporal_store(v, reinterpret_cast<ew_f4 *>(p));
#else
  *reinterpret_cast<float4 *>(p) = o;
#endif
}

// Large matrices: the 16-byte elements of all rows are dealt out to the threads as one flat sequence (no idle lanes
// whatever the row length -- 1320 columns fill 330 of a row's 512 lane slots in the kernel above), and a thread has
// UNROLL elements' loads in flight before it computes and stores.
template <class Op, int UNROLL>
__global__ __launch_bounds__(256) void ew_flat_kernel(EwArgs A, MatrixDim d, unsigned c4_per_row, unsigned total4) {
  const unsigned nthreads = gridDim.x * 256;
  unsigned e = blockIdx.x * 256 + threadIdx.x;
  for (; e < total4; e += nthreads * UNROLL) {
    float4 m[UNROLL], a[UNROLL], b[UNROLL], cv[UNROLL];
    float rv[UNROLL];
    size_t idx[UNROLL];
    bool on[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const unsigned eu = e + u * nthreads;
      on[u] = eu < total4;
      const unsigned r = on[u] ? eu / c4_per_row : 0, c = on[u] ? (eu - r * c4_per_row) * 4 : 0;
      idx[u] = (size_t)r * d.stride + c;
      if (Op::kReadDst) m[u] = *reinterpret_cast<const float4 *>(A.dst + idx[u]);
      if (Op::kUseA) a[u] = ld_stream(A.a + idx[u]);
      if (Op::kUseB) b[u] = ld_stream(A.b + idx[u]);
      if (Op::kColV) cv[u] = *reinterpret_cast<const float4 *>(A.colv + c);
      rv[u] = Op::kRowV ? A.rowv[r] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      if (!on[u]) continue;
      float4 o;
      o.x = Op::apply(Op::kReadDst ? m[u].x : 0.f, Op::kUseA ? a[u].x : 0.f, Op::kUseB ? b[u].x : 0.f, Op::kColV ? cv[u].x : 0.f, rv[u], A.p0, A.p1);
      o.y = Op::apply(Op::kReadDst ? m[u].y : 0.f, Op::kUseA ? a[u].y : 0.f, Op::kUseB ? b[u].y : 0.f, Op::kColV ? cv[u].y : 0.f, rv[u], A.p0, A.p1);
      o.z = Op::apply(Op::kReadDst ? m[u].z : 0.f, Op::kUseA ? a[u].z : 0.f, Op::kUseB ? b[u].z : 0.f, Op::kColV ? cv[u].z : 0.f, rv[u], A.p0, A.p1);
      o.w = Op::apply(Op::kReadDst ? m[u].w : 0.f, Op::kUseA ? a[u].w : 0.f, Op::kUseB ? b[u].w : 0.f, Op::kColV ? cv[u].w : 0.f, rv[u], A.p0, A.p1);
      if (Op::kReadDst) *reinterpret_cast<float4 *>(A.dst + idx[u]) = o;
      else st_stream(A.dst + idx[u], o);
    }
  }
}

template <class Op>
int launch_ew(khStream_t s, EwArgs A, MatrixDim d) {
  KH_CHECK_INIT();
  KH_REQUIRE(kh::dim_ok(d));
  if (d.rows == 0 || d.cols == 0) return KH_OK;
  KH_REQUIRE(A.dst != nullptr);
  KH_REQUIRE(!Op::kUseA || A.a != nullptr);
  KH_REQUIRE(!Op::kUseB || A.b != nullptr);
  KH_REQUIRE(!Op::kColV || A.colv != nullptr);
  KH_REQUIRE(!Op::kRowV || A.rowv != nullptr);
  const bool vec = (d.cols % 4 == 0) && (d.stride % 4 == 0) && kh::aligned16(A.dst) &&
                   (!Op::kUseA || kh::aligned16(A.a)) && (!Op::kUseB || kh::aligned16(A.b)) &&
                   (!Op::kColV || kh::aligned16(A.colv));
  hipStream_t st = kh::pick(s);
  const unsigned long long total4 = (unsigned long long)d.rows * (d.cols / 4);
  if (vec && total4 >= (1ull << 20) && total4 < (1ull << 31)) {
    constexpr int kUnroll = 4;
    // as many workgroups as one pass takes (measured at 32768 x 1320: 16384 workgroups 61 us, 8192 with a loop 64, 2048 70)
    unsigned blocks = (unsigned)((total4 + 256 * kUnroll - 1) / (256 * kUnroll));
    if (blocks > 65536u) blocks = 65536u;
    hipLaunchKernelGGL((ew_flat_kernel<Op, kUnroll>), dim3(blocks), dim3(256), 0, st, A, d, (unsigned)(d.cols / 4), (unsigned)total4);
    return kh::launch_status();
  }
  const int per = vec ? 4 : 1;
  const int gx = (d.cols + 256 * per - 1) / (256 * per);
  // enough row-blocks to give every CU several waves, never more rows than exist
  int gy = d.rows;
  const int cap = 8192 / (gx > 0 ? gx : 1);
  if (gy > cap) gy = cap < 1 ? 1 : cap;
  dim3 grid(gx, gy), block(256);
  if (vec)
    hipLaunchKernelGGL((ew_kernel<Op, 4>), grid, block, 0, st, A, d);
  else
    hipLaunchKernelGGL((ew_kernel<Op, 1>), grid, block, 0, st, A, d);
  return kh::launch_status();
}

EwArgs args(float *dst, const float *a = nullptr, const float *b = nullptr, const float *colv = nullptr,
            const float *rowv = nullptr, float p0 = 0.f, float p1 = 0.f) {
  EwArgs A;
  A.dst = dst;
  A.a = a;
  A.b = b;
  A.colv = colv;
  A.rowv = rowv;
  A.p0 = p0;
  A.p1 = p1;
  return A;
}

// ---- the few kernels that do not fit the streaming template --------------------------------

__global__ void set_const_i32_kernel(int32_t *mat, int32_t value, MatrixDim d) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= d.cols) return;
  for (int r = blockIdx.y; r < d.rows; r += gridDim.y) mat[(size_t)r * d.stride + c] = value;
}

__global__ void part_set_const_kernel(float *mat, float value, MatrixDim d, int ro, int r, int co, int c) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;  // column inside the block
  if (i >= c) return;
  const int col = co + i;
  if (col >= d.cols) return;
  for (int j = blockIdx.y; j < r; j += gridDim.y) {
    const int row = ro + j;
    if (row < d.rows) mat[(size_t)row * d.stride + col] = value;
  }
}

__global__ void part_add_mat_kernel(float alpha, const float *A, MatrixDim da, int ro, int co, float beta,
                                    float *dst, MatrixDim d) {
  const int ai = blockIdx.x * blockDim.x + threadIdx.x;
  if (ai >= da.cols) return;
  const int i = ai + co;
  if (i < 0 || i >= d.cols) return;
  for (int aj = blockIdx.y; aj < da.rows; aj += gridDim.y) {
    const int j = aj + ro;
    if (j < 0 || j >= d.rows) continue;
    const size_t idx = (size_t)j * d.stride + i;
    dst[idx] = alpha * A[(size_t)aj * da.stride + ai] + beta * dst[idx];
  }
}

__global__ void add_vec_to_partial_rows_kernel(float alpha, int offset, const float *row, int dim, float beta,
                                               float *dst, MatrixDim d) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= dim) return;
  const int i = v + offset;
  if (i < 0 || i >= d.cols) return;
  const float rv = alpha * row[v];
  for (int j = blockIdx.y; j < d.rows; j += gridDim.y) {
    const size_t idx = (size_t)j * d.stride + i;
    dst[idx] = rv + beta * dst[idx];
  }
}

__global__ void regularize_l1_kernel(float *wei, float *grad, float l1, float lr, MatrixDim d) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= d.cols) return;
  for (int j = blockIdx.y; j < d.rows; j += gridDim.y) {
    const size_t idx = (size_t)j * d.stride + i;
    const float w = wei[idx];
    if (w == 0.0f) continue;  // zero weights stay zero
    const float l1_signed = (w < 0.0f) ? -l1 : l1;
    const float after = w - lr * grad[idx] - l1_signed;  // simulated update
    if ((after > 0.0f) ^ (w > 0.0f)) {
      wei[idx] = 0.0f;
      grad[idx] = 0.0f;
    } else {
      wei[idx] = w - l1_signed;
    }
  }
}

// One wave per row: max |x|, then the scaled sum of squares (double), then the rescale.
__global__ __launch_bounds__(256) void row_l2upperbound_kernel(float *mat, float bound, MatrixDim d) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= d.rows) return;
  float *p = mat + (size_t)row * d.stride;
  float mx = 0.f;
  for (int c = lane; c < d.cols; c += 64) mx = fmaxf(mx, fabsf(p[c]));
  mx = kh::wave_max(mx);
  const double max_mag = mx;
  double acc = 0.0;
  for (int c = lane; c < d.cols; c += 64) {
    double t = (double)p[c] / max_mag;
    acc += t * t;
  }
  acc = kh::wave_sum_d(acc);
  const double l2norm = sqrt(acc);
  if (max_mag * l2norm > (double)bound) {
    for (int c = lane; c < d.cols; c += 64) p[c] = (float)((((double)p[c] / max_mag) / l2norm) * (double)bound);
  }
}

// Counts the NaN and the +infinity elements of a matrix: counts[0] += #NaN, counts[1] += #(+inf).  A lane keeps
// its own tallies, a wave adds them up with one ballot-free shuffle tree and only a wave that found something
// touches the counters, so a clean matrix costs one streaming read and no atomics.
__global__ void count_nonfinite_kernel(const float *mat, MatrixDim d, int *counts) {
  int n_nan = 0, n_inf = 0;
  const float inf = __builtin_inff();
  const bool vec = (d.stride % 4 == 0) && ((reinterpret_cast<size_t>(mat) & 15) == 0);
  for (int r = blockIdx.x; r < d.rows; r += gridDim.x) {
    const float *row = mat + (size_t)r * d.stride;
    int c0 = 0;
    if (vec) {
      const int nv = d.cols / 4;
      for (int v = threadIdx.x; v < nv; v += blockDim.x) {
        const float4 x = reinterpret_cast<const float4 *>(row)[v];
        n_nan += (x.x != x.x) + (x.y != x.y) + (x.z != x.z) + (x.w != x.w);
        n_inf += (x.x == inf) + (x.y == inf) + (x.z == inf) + (x.w == inf);
      }
      c0 = nv * 4;
    }
    for (int c = c0 + threadIdx.x; c < d.cols; c += blockDim.x) {
      const float x = row[c];
      n_nan += (x != x);
      n_inf += (x == inf);
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    n_nan += __shfl_down(n_nan, off, 64);
    n_inf += __shfl_down(n_inf, off, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    if (n_nan) atomicAdd(&counts[0], n_nan);
    if (n_inf) atomicAdd(&counts[1], n_inf);
  }
}

dim3 grid2d(int cols, int rows) {
  int gx = (cols + 255) / 256;
  int gy = rows < 1 ? 1 : rows;
  int cap = 8192 / (gx > 0 ? gx : 1);
  if (gy > cap) gy = cap < 1 ? 1 : cap;
  return dim3(gx > 0 ? gx : 1, gy);
}

}  // namespace

KH_EXPORT int hipI32_set_const(khStream_t s, int32_t *mat, int32_t value, MatrixDim d) {
  KH_CHECK_INIT();
  KH_REQUIRE(kh::dim_ok(d));
  if (d.rows == 0 || d.cols == 0) return KH_OK;
  KH_REQUIRE(mat != nullptr);
  hipLaunchKernelGGL(set_const_i32_kernel, grid2d(d.cols, d.rows), dim3(256), 0, kh::pick(s), mat, value, d);
  return kh::launch_status();
}

KH_EXPORT int hipF_count_nonfinite(khStream_t s, const float *mat, MatrixDim d, int32_t *counts) {
  KH_CHECK_INIT();
  KH_REQUIRE(kh::dim_ok(d) && counts != nullptr);
  hipStream_t st = kh::pick(s);
  KH_HIP(hipMemsetAsync(counts, 0, 2 * sizeof(int32_t), st));
  if (d.rows == 0 || d.cols == 0) return KH_OK;
  KH_REQUIRE(mat != nullptr);
  const int blocks = d.rows < 2048 ? d.rows : 2048;
  hipLaunchKernelGGL(count_nonfinite_kernel, dim3(blocks), dim3(256), 0, st, mat, d, counts);
  return kh::launch_status();
}

KH_EXPORT int hipF_set_const(khStream_t s, float *mat, float value, MatrixDim d) {
  return launch_ew<OpSetConst>(s, args(mat, 0, 0, 0, 0, value), d);
}
KH_EXPORT int hipF_part_set_const(khStream_t s, float *mat, float value, MatrixDim d, int32_t ro, int32_t r,
                                  int32_t co, int32_t c) {
  KH_CHECK_INIT();
  KH_REQUIRE(kh::dim_ok(d) && ro >= 0 && co >= 0 && r >= 0 && c >= 0);
  if (d.rows == 0 || d.cols == 0 || r == 0 || c == 0) return KH_OK;
  KH_REQUIRE(mat != nullptr);
  hipLaunchKernelGGL(part_set_const_kernel, grid2d(c, r), dim3(256), 0, kh::pick(s), mat, value, d, ro, r, co, c);
  return kh::launch_status();
}
KH_EXPORT int hipF_binarize(khStream_t s, float *mat, float thres, MatrixDim d) {
  return launch_ew<OpBinarize>(s, args(mat, 0, 0, 0, 0, thres), d);
}
KH_EXPORT int hipF_add_const(khStream_t s, float *mat, float value, MatrixDim d) {
  return launch_ew<OpAddConst>(s, args(mat, 0, 0, 0, 0, value), d);
}
KH_EXPORT int hipF_power(khStream_t s, float *mat, float pw, MatrixDim d) {
  return launch_ew<OpPower>(s, args(mat, 0, 0, 0, 0, pw), d);
}
KH_EXPORT int hipF_round(khStream_t s, float *mat, MatrixDim d) { return launch_ew<OpRound>(s, args(mat), d); }
KH_EXPORT int hipF_scale(khStream_t s, float *mat, float value, MatrixDim d) {
  return launch_ew<OpScale>(s, args(mat, 0, 0, 0, 0, value), d);
}
KH_EXPORT int hipF_apply_floor(khStream_t s, float *mat, float value, MatrixDim d) {
  return launch_ew<OpFloor>(s, args(mat, 0, 0, 0, 0, value), d);
}
KH_EXPORT int hipF_apply_truncate(khStream_t s, float *mat, float low, float high, MatrixDim d) {
  return launch_ew<OpTruncate>(s, args(mat, 0, 0, 0, 0, low, high), d);
}
KH_EXPORT int hipF_apply_log(khStream_t s, float *mat, MatrixDim d) { return launch_ew<OpLog>(s, args(mat), d); }
KH_EXPORT int hipF_apply_exp(khStream_t s, float *mat, MatrixDim d) { return launch_ew<OpExp>(s, args(mat), d); }
KH_EXPORT int hipF_invert_elements(khStream_t s, float *data, MatrixDim d) {
  return launch_ew<OpInvert>(s, args(data), d);
}
KH_EXPORT int hipF_mul_elements(khStream_t s, float *mat, const float *A, MatrixDim d) {
  return launch_ew<OpMulElements>(s, args(mat, A), d);
}
KH_EXPORT int hipF_div_elements(khStream_t s, float *mat, const float *A, MatrixDim d) {
  return launch_ew<OpDivElements>(s, args(mat, A), d);
}
KH_EXPORT int hipF_mul_cols_vec(khStream_t s, float *mat, const float *scale, MatrixDim d) {
  return launch_ew<OpMulColsVec>(s, args(mat, 0, 0, scale), d);
}
KH_EXPORT int hipF_mul_rows_vec(khStream_t s, float *mat, const float *scale, MatrixDim d) {
  return launch_ew<OpMulRowsVec>(s, args(mat, 0, 0, 0, scale), d);
}
KH_EXPORT int hipF_div_cols_vec(khStream_t s, float *mat, const float *vec_div, MatrixDim d) {
  return launch_ew<OpDivColsVec>(s, args(mat, 0, 0, vec_div), d);
}
KH_EXPORT int hipF_div_rows_vec(khStream_t s, float *mat, const float *vec_div, MatrixDim d) {
  return launch_ew<OpDivRowsVec>(s, args(mat, 0, 0, 0, vec_div), d);
}
KH_EXPORT int hipF_add_mat(khStream_t s, float alpha, const float *A, float beta, float *dst, MatrixDim d) {
  if (beta == 0.0f) return launch_ew<OpAddMatBeta0>(s, args(dst, A, 0, 0, 0, alpha, beta), d);
  return launch_ew<OpAddMat>(s, args(dst, A, 0, 0, 0, alpha, beta), d);
}
KH_EXPORT int hipF_part_add_mat(khStream_t s, float alpha, const float *A, MatrixDim da, int32_t ro, int32_t co,
                                float beta, float *dst, MatrixDim d) {
  KH_CHECK_INIT();
  KH_REQUIRE(kh::dim_ok(d) && kh::dim_ok(da));
  if (d.rows == 0 || d.cols == 0 || da.rows == 0 || da.cols == 0) return KH_OK;
  KH_REQUIRE(A != nullptr && dst != nullptr);
  hipLaunchKernelGGL(part_add_mat_kernel, grid2d(da.cols, da.rows), dim3(256), 0, kh::pick(s), alpha, A, da, ro,
                     co, beta, dst, d);
  return kh::launch_status();
}
KH_EXPORT int hipF_add_vec_to_cols(khStream_t s, float alpha, const float *col, float beta, float *dst,
                                   MatrixDim d) {
  if (beta == 0.0f) return launch_ew<OpAddVecToColsBeta0>(s, args(dst, 0, 0, 0, col, alpha, beta), d);
  return launch_ew<OpAddVecToCols>(s, args(dst, 0, 0, 0, col, alpha, beta), d);
}
KH_EXPORT int hipF_add_vec_to_rows(khStream_t s, float alpha, const float *row, float beta, float *dst,
                                   MatrixDim d) {
  if (beta == 0.0f) return launch_ew<OpAddVecToRowsBeta0>(s, args(dst, 0, 0, row, 0, alpha, beta), d);
  return launch_ew<OpAddVecToRows>(s, args(dst, 0, 0, row, 0, alpha, beta), d);
}
KH_EXPORT int hipF_add_vec_to_partial_rows(khStream_t s, float alpha, int32_t offset, const float *row,
                                           int32_t dim, float beta, float *dst, MatrixDim d) {
  KH_CHECK_INIT();
  KH_REQUIRE(kh::dim_ok(d) && dim >= 0);
  if (d.rows == 0 || d.cols == 0 || dim == 0) return KH_OK;
  KH_REQUIRE(row != nullptr && dst != nullptr);
  hipLaunchKernelGGL(add_vec_to_partial_rows_kernel, grid2d(dim, d.rows), dim3(256), 0, kh::pick(s), alpha,
                     offset, row, dim, beta, dst, d);
  return kh::launch_status();
}
KH_EXPORT int hipF_log_add_exp_mat(khStream_t s, const float *A, float *dst, MatrixDim d) {
  return launch_ew<OpLogAddExp>(s, args(dst, A), d);
}
KH_EXPORT int hipF_apply_row_l2upperbound(khStream_t s, float *mat, float bound, MatrixDim d) {
  KH_CHECK_INIT();
  KH_REQUIRE(kh::dim_ok(d));
  if (d.rows == 0 || d.cols == 0) return KH_OK;
  KH_REQUIRE(mat != nullptr);
  hipLaunchKernelGGL(row_l2upperbound_kernel, dim3((d.rows + 3) / 4), dim3(256), 0, kh::pick(s), mat, bound, d);
  return kh::launch_status();
}
KH_EXPORT int hipF_relu(khStream_t s, float *y, const float *x, MatrixDim d) {
  return launch_ew<OpRelu>(s, args(y, x), d);
}
KH_EXPORT int hipF_diff_relu(khStream_t s, float *eout, const float *e, const float *y, MatrixDim d) {
  return launch_ew<OpDiffRelu>(s, args(eout, e, y), d);
}
KH_EXPORT int hipF_softrelu(khStream_t s, float *y, const float *x, MatrixDim d) {
  return launch_ew<OpSoftRelu>(s, args(y, x), d);
}
KH_EXPORT int hipF_diff_softrelu(khStream_t s, float *eout, const float *e, const float *x, MatrixDim d) {
  return launch_ew<OpDiffSoftRelu>(s, args(eout, e, x), d);
}
KH_EXPORT int hipF_sigmoid(khStream_t s, float *y, const float *x, MatrixDim d) {
  return launch_ew<OpSigmoid>(s, args(y, x), d);
}
KH_EXPORT int hipF_diff_sigmoid(khStream_t s, float *eout, const float *e, const float *y, MatrixDim d) {
  return launch_ew<OpDiffSigmoid>(s, args(eout, e, y), d);
}
KH_EXPORT int hipF_regularize_l1(khStream_t s, float *wei, float *grad, float l1, float lr, MatrixDim d) {
  KH_CHECK_INIT();
  KH_REQUIRE(kh::dim_ok(d));
  if (d.rows == 0 || d.cols == 0) return KH_OK;
  KH_REQUIRE(wei != nullptr && grad != nullptr);
  hipLaunchKernelGGL(regularize_l1_kernel, grid2d(d.cols, d.rows), dim3(256), 0, kh::pick(s), wei, grad, l1, lr, d);
  return kh::launch_status();
}
KH_EXPORT int hipF_binarize_probs(khStream_t s, float *states, const float *probs, const float *rand, MatrixDim d) {
  return launch_ew<OpBinarizeProbs>(s, args(states, probs, rand), d);
}
