// Row / column reductions of the hot path: bias-gradient column sums, row sums, row softmax,
// row argmax, cross-entropy and MSE evaluation.  All HBM-bound: one pass over the matrix,
// 64-lane wavefront reductions (DPP/shuffle), LDS only to combine the waves of a block.
#include "common.h"

// Scratch of hipF_xent_eval_static, the entry point that allocates nothing: the arrival counter of the
// "last workgroup sums up" step (first 256 bytes) followed by the per-row results nobody asked for
// (log-posterior and arg-max of kStaticXentRows rows).  Every GPU the library runs on holds its own
// copy until the process ends; host code never passes the array's name to a kernel, only the
// device address hipGetSymbolAddress returned for the current device.
constexpr int kStaticXentRows = 65536;
constexpr size_t kStaticTicketBytes = 256;
static __device__ __align__(256) unsigned char scratch[kStaticTicketBytes + (size_t)kStaticXentRows * 8];

namespace {
constexpr int kMaxDevices = 64;
unsigned char *g_scratch_ptr[kMaxDevices] = {nullptr};  // device address of `scratch`, per device ordinal
}

namespace kh {
// Called by hipK_init with `device` current: looks the array's address up once per device.
int static_scratch_setup(int device) {
  if (device < 0 || device >= kMaxDevices) return KH_ERR_INVALID_ARG;
  if (g_scratch_ptr[device]) return KH_OK;
  unsigned char *ptr = nullptr;
  KH_HIP(hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(scratch)));
  if (!ptr) return KH_ERR_NOT_INITIALISED;
  g_scratch_ptr[device] = ptr;
  return KH_OK;
}
}  // namespace kh

namespace {

// 16-byte buffer loads from a row of floats: the descriptor ends with the group of 4 that holds the row's last column,
// so lanes past the end read 0 without a branch around the access (a ragged last group reads up to 3 floats of the row's
// pitch behind the last column; the kernels mask them out).
typedef int kh_v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t row_rsrc(const float *p, int cols) {
  return __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, ((cols + 3) & ~3) * 4, 0x00020000);
}
__device__ __forceinline__ float4 row_load4(__amdgpu_buffer_rsrc_t rs, int c) {
  const kh_v4i r = __builtin_amdgcn_raw_buffer_load_b128(rs, c * 4, 0, 0);
  return make_float4(__int_as_float(r.x), __int_as_float(r.y), __int_as_float(r.z), __int_as_float(r.w));
}
// Stores to a row write the row's columns and nothing else -- a caller's matrix may be a window of a larger one, whose
// row pitch then holds live data behind the last column.  Whole groups of 4 go out as one 16-byte access bounded by a
// descriptor that ends with the last whole group; the lane that holds a ragged last group (a row length that is not a
// multiple of 4) writes its columns one by one.
struct RowOut {
  __amdgpu_buffer_rsrc_t whole, exact;
  int cols;
};
__device__ __forceinline__ RowOut row_out(float *p, int cols) {
  RowOut o;
  o.whole = __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, (cols & ~3) * 4, 0x00020000);
  o.exact = __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, cols * 4, 0x00020000);
  o.cols = cols;
  return o;
}
__device__ __forceinline__ void row_store4(const RowOut &o, int c, float a, float b, float cc, float d) {
  kh_v4i v = {__float_as_int(a), __float_as_int(b), __float_as_int(cc), __float_as_int(d)};
  __builtin_amdgcn_raw_buffer_store_b128(v, o.whole, c * 4, 0, 0);
  if (c < o.cols && c + 4 > o.cols) {  // (at most one lane of a row, and none when the length is a multiple of 4)
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(a), o.exact, c * 4, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(b), o.exact, c * 4 + 4, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(cc), o.exact, c * 4 + 8, 0, 0);
  }
}
// ---- vec[c] = alpha * sum_r mat[r,c] + beta * vec[c] ---------------------------------------
// Block = 64 columns x 16 row groups; lane -> column keeps every wave load a contiguous 256 B.
// The per-group partial sums are combined in a fixed order, so results are reproducible.
__global__ __launch_bounds__(1024) void col_sums_kernel(float alpha, const float *mat, MatrixDim d, float beta,
                                                        float *vec) {
  __shared__ float part[16][64];
  const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  float acc = 0.f;
  if (c < d.cols) {
    const float *p = mat + c;
    int r = g;
    // four independent loads in flight per lane
    for (; r + 48 < d.rows; r += 64) {
      float a0 = p[(size_t)r * d.stride], a1 = p[(size_t)(r + 16) * d.stride];
      float a2 = p[(size_t)(r + 32) * d.stride], a3 = p[(size_t)(r + 48) * d.stride];
      acc += (a0 + a1) + (a2 + a3);
    }
    for (; r < d.rows; r += 16) acc += p[(size_t)r * d.stride];
  }
  part[g][lane] = acc;
  __syncthreads();
  if (g == 0 && c < d.cols) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += part[i][lane];
    vec[c] = (beta == 0.f) ? alpha * t : alpha * t + beta * vec[c];
  }
}

// Tall matrices (a whole frame cache): the same sums in two stages, so that every compute unit reads.  Stage 1: the
// rows are cut into `nslab` slabs; a 256-thread workgroup sums one slab of up to 2048 columns, its 4 waves taking
// rows in turn, combined in wave order through LDS into part[slab][col].  Stage 2 is the single-stage kernel above on
// the matrix of partial sums.  The order of the additions is fixed by the shape alone: same bits every run.
template <int NV>
__global__ __launch_bounds__(1024) void col_sums_slab_kernel(const float *mat, MatrixDim d, int rows_per_slab, float *part,
                                                             int ldpart) {
  // this workgroup: rows [r0, r1) x columns [c0, c0 + 256 * NV); a lane owns 4 adjacent columns in each of NV groups
  // of 256, so a wave instruction reads 1 KB of one row and each of the 16 waves walks along whole rows (two in
  // flight); the waves' sums are combined four at a time through LDS, in wave order
  __shared__ float red[4][256 * NV];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c0 = blockIdx.x * 256 * NV;
  const int r0 = blockIdx.y * rows_per_slab, r1 = min(d.rows, r0 + rows_per_slab);
  const int ncols = min(d.cols - c0, 256 * NV);
  float4 acc[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  int r = r0 + w;
  for (; r + 16 < r1; r += 32) {
    const __amdgpu_buffer_rsrc_t ra = row_rsrc(mat + (size_t)r * d.stride + c0, ncols);
    const __amdgpu_buffer_rsrc_t rb = row_rsrc(mat + (size_t)(r + 16) * d.stride + c0, ncols);
    float4 xa[NV], xb[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) xa[k] = row_load4(ra, (lane + k * 64) * 4);
#pragma unroll
    for (int k = 0; k < NV; ++k) xb[k] = row_load4(rb, (lane + k * 64) * 4);
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      acc[k].x += xa[k].x; acc[k].y += xa[k].y; acc[k].z += xa[k].z; acc[k].w += xa[k].w;
    }
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      acc[k].x += xb[k].x; acc[k].y += xb[k].y; acc[k].z += xb[k].z; acc[k].w += xb[k].w;
    }
  }
  for (; r < r1; r += 16) {
    const __amdgpu_buffer_rsrc_t ra = row_rsrc(mat + (size_t)r * d.stride + c0, ncols);
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const float4 x = row_load4(ra, (lane + k * 64) * 4);
      acc[k].x += x.x; acc[k].y += x.y; acc[k].z += x.z; acc[k].w += x.w;
    }
  }
  float tot[NV];  // thread t (< 256) ends up with columns t, t + 256, ...
#pragma unroll
  for (int k = 0; k < NV; ++k) tot[k] = 0.f;
  for (int round = 0; round < 4; ++round) {
    __syncthreads();
    if ((w >> 2) == round) {
#pragma unroll
      for (int k = 0; k < NV; ++k) *reinterpret_cast<float4 *>(&red[w & 3][(lane + k * 64) * 4]) = acc[k];
    }
    __syncthreads();
    if (threadIdx.x < 256) {
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        const int i = threadIdx.x + k * 256;
        tot[k] += (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
      }
    }
  }
  if (threadIdx.x < 256) {
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int i = threadIdx.x + k * 256;
      if (i < ncols) part[(size_t)blockIdx.y * ldpart + c0 + i] = tot[k];
    }
  }
}
// any layout (rows not 16-byte aligned): a lane owns one column of 64
__global__ __launch_bounds__(256) void col_sums_slab_scalar_kernel(const float *mat, MatrixDim d, int rows_per_slab, float *part,
                                                                   int ldpart) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  const int r0 = blockIdx.y * rows_per_slab, r1 = min(d.rows, r0 + rows_per_slab);
  float acc = 0.f;
  if (c < d.cols)
    for (int r = r0 + w; r < r1; r += 4) acc += mat[(size_t)r * d.stride + c];
  red[w][lane] = acc;
  __syncthreads();
  if (w == 0 && c < d.cols) part[(size_t)blockIdx.y * ldpart + c] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}
// ---- vec[r] = alpha * sum_c mat[r,c] + beta * vec[r] ---------------------------------------
__global__ __launch_bounds__(256) void row_sums_kernel(float alpha, const float *mat, MatrixDim d, float beta,
                                                       float *vec) {
  __shared__ float red[16];
  const int r = blockIdx.x;
  const float *p = mat + (size_t)r * d.stride;
  float acc = 0.f;
  for (int c = threadIdx.x; c < d.cols; c += 256) acc += p[c];
  const float t = kh::block_sum(acc, red);
  if (threadIdx.x == 0) vec[r] = (beta == 0.f) ? alpha * t : alpha * t + beta * vec[r];
}

// ---- row softmax ---------------------------------------------------------------------------
// One block per row.  The row is staged in LDS while its maximum is found, exponentiated in
// place while the sum is formed, then scaled on the way out: x is read once, y written once.
// Rows longer than the LDS staging buffer re-read x from L2 instead.
#ifndef KH_SOFTMAX_THREADS
#define KH_SOFTMAX_THREADS 512
#endif
constexpr int kSoftmaxThreads = KH_SOFTMAX_THREADS;
constexpr int kSoftmaxLdsFloats = 12288;  // 48 KiB: up to 3 blocks per CU

// x may be given as `nslab` partial slabs (split-K GEMM output) plus a per-column bias; the
// row value is then bias[c] + slab0[c] + slab1[c] + ... summed in slab order.
template <bool STAGED>
__global__ __launch_bounds__(kSoftmaxThreads) void softmax_kernel(float *y, int ldy, const float *x, int ldx,
                                                                  int nslab, size_t slab_stride, const float *bias,
                                                                  int cols) {
  __shared__ float row[STAGED ? kSoftmaxLdsFloats : 1];
  __shared__ float red[16];
  const int r = blockIdx.x;
  const float *px = x + (size_t)r * ldx;
  float *py = y + (size_t)r * ldy;
  const int t = threadIdx.x;

  auto value = [&](int c) {
    float v = px[c];
    for (int s = 1; s < nslab; ++s) v += px[(size_t)s * slab_stride + c];
    if (bias) v += bias[c];
    return v;
  };

  float mx = -INFINITY;
  for (int c = t; c < cols; c += kSoftmaxThreads) {
    const float v = value(c);
    if (STAGED) row[c] = v; else py[c] = v;
    mx = fmaxf(mx, v);
  }
  mx = kh::block_max(mx, red);

  float sum = 0.f;
  for (int c = t; c < cols; c += kSoftmaxThreads) {
    const float e = expf((STAGED ? row[c] : py[c]) - mx);
    if (STAGED) row[c] = e; else py[c] = e;
    sum += e;
  }
  sum = kh::block_sum(sum, red);
  const float inv = 1.0f / sum;
  for (int c = t; c < cols; c += kSoftmaxThreads) py[c] = (STAGED ? row[c] : py[c]) * inv;
}

// Register-resident form: every thread owns NV float4 of the row (slabs and bias summed in the same
// order as above), all loads in flight together, one read and one write of the row.  Needs 16-byte
// aligned rows / slabs / bias and cols <= 4 * kSoftmaxThreads * NV.
// The row in registers: v[] receives exp(x - max) of this thread's columns (0 past the end), the
// return value is 1 / sum.
// All accesses are 16-byte buffer loads / stores whose descriptor ends with the row's last whole group of 4
// columns (rows are padded to a multiple of 4 floats; a 16-byte access that crosses the descriptor's end is out
// of range as a whole): lanes past the end read 0 and write nothing, so there is no branch around any access
// and the loads of a phase -- every slab and the bias -- are all in flight together.  (With a branch per
// access, every guarded load became a memory round trip of its own.)  Columns of the last group that lie past
// `cols` hold padding: they are masked out of max and sum and stored as 0.
template <int NV>
__device__ __forceinline__ float softmax_row_regs(float4 (&v)[NV], const float *px, int nslab, size_t slab_stride,
                                                  const float *bias, int cols, float *red) {
  const int t = threadIdx.x;
  const __amdgpu_buffer_rsrc_t rs0 = row_rsrc(px, cols);
#pragma unroll
  for (int k = 0; k < NV; ++k) v[k] = row_load4(rs0, (t + k * kSoftmaxThreads) * 4);
  float4 b4[NV];
  if (bias) {
    // the bias vector is not padded: whole groups of 4 by the descriptor, a ragged end element by element
    const int whole = cols & ~3;
    const __amdgpu_buffer_rsrc_t rsb = row_rsrc(bias, whole);
#pragma unroll
    for (int k = 0; k < NV; ++k) b4[k] = row_load4(rsb, (t + k * kSoftmaxThreads) * 4);
    if (cols != whole) {
#pragma unroll
      for (int k = 0; k < NV; ++k)
        if ((t + k * kSoftmaxThreads) * 4 == whole) {
          b4[k].x = bias[whole];
          if (whole + 1 < cols) b4[k].y = bias[whole + 1];
          if (whole + 2 < cols) b4[k].z = bias[whole + 2];
        }
    }
  }
  for (int sl = 1; sl < nslab; ++sl) {  // slabs in order, then the bias: the order of the sums fixes the bits
    const __amdgpu_buffer_rsrc_t rs = row_rsrc(px + (size_t)sl * slab_stride, cols);
    float4 a[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) a[k] = row_load4(rs, (t + k * kSoftmaxThreads) * 4);
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      v[k].x += a[k].x; v[k].y += a[k].y; v[k].z += a[k].z; v[k].w += a[k].w;
    }
  }
  if (bias) {
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      v[k].x += b4[k].x; v[k].y += b4[k].y; v[k].z += b4[k].z; v[k].w += b4[k].w;
    }
  }
  float mx = -INFINITY;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int c = (t + k * kSoftmaxThreads) * 4;
    mx = fmaxf(mx, (c < cols) ? v[k].x : -INFINITY);
    mx = fmaxf(mx, (c + 1 < cols) ? v[k].y : -INFINITY);
    mx = fmaxf(mx, (c + 2 < cols) ? v[k].z : -INFINITY);
    mx = fmaxf(mx, (c + 3 < cols) ? v[k].w : -INFINITY);
  }
  mx = kh::block_max(mx, red);
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int c = (t + k * kSoftmaxThreads) * 4;
    v[k].x = (c < cols) ? expf(v[k].x - mx) : 0.f;
    v[k].y = (c + 1 < cols) ? expf(v[k].y - mx) : 0.f;
    v[k].z = (c + 2 < cols) ? expf(v[k].z - mx) : 0.f;
    v[k].w = (c + 3 < cols) ? expf(v[k].w - mx) : 0.f;
    sum += (v[k].x + v[k].y) + (v[k].z + v[k].w);
  }
  sum = kh::block_sum(sum, red);
  return 1.0f / sum;
}
template <int NV>
__global__ __launch_bounds__(kSoftmaxThreads) void softmax_vec_kernel(float *y, int ldy, const float *x, int ldx,
                                                                      int nslab, size_t slab_stride,
                                                                      const float *bias, int cols) {
  __shared__ float red[16];
  const int r = blockIdx.x;
  float *py = y + (size_t)r * ldy;
  float4 v[NV];
  const float inv = softmax_row_regs<NV>(v, x + (size_t)r * ldx, nslab, slab_stride, bias, cols, red);
  const RowOut rsy = row_out(py, cols);
#pragma unroll
  for (int k = 0; k < NV; ++k)
    row_store4(rsy, (threadIdx.x + k * kSoftmaxThreads) * 4, v[k].x * inv, v[k].y * inv, v[k].z * inv, v[k].w * inv);
}

// Short rows (cols <= 256 * NV): one WAVE per row, four rows per workgroup.  The row lives in NV float4 per lane
// (16-byte buffer loads bounded by the row's descriptor, all in flight together: every slab and the bias, summed in
// slab order and then the bias, as in softmax_row_regs), maximum and sum are wave reductions: no LDS, no barrier.
// v[] receives exp(x - max) of this lane's columns (0 past the end), the return value is 1 / sum.
template <int NV>
__device__ __forceinline__ float softmax_row_wave(float4 (&v)[NV], const float *px, int nslab, size_t slab_stride,
                                                  const float *bias, int cols) {
  const int lane = threadIdx.x & 63;
  const __amdgpu_buffer_rsrc_t rs0 = row_rsrc(px, cols);
#pragma unroll
  for (int k = 0; k < NV; ++k) v[k] = row_load4(rs0, (lane + k * 64) * 4);
  float4 b4[NV];
  if (bias) {
    const int whole = cols & ~3;  // (the bias vector is not padded: whole groups of 4 by the descriptor, a ragged end by hand)
    const __amdgpu_buffer_rsrc_t rsb = row_rsrc(bias, whole);
#pragma unroll
    for (int k = 0; k < NV; ++k) b4[k] = row_load4(rsb, (lane + k * 64) * 4);
    if (cols != whole) {
#pragma unroll
      for (int k = 0; k < NV; ++k)
        if ((lane + k * 64) * 4 == whole) {
          b4[k].x = bias[whole];
          if (whole + 1 < cols) b4[k].y = bias[whole + 1];
          if (whole + 2 < cols) b4[k].z = bias[whole + 2];
        }
    }
  }
  for (int sl = 1; sl < nslab; ++sl) {
    const __amdgpu_buffer_rsrc_t rs = row_rsrc(px + (size_t)sl * slab_stride, cols);
    float4 a[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) a[k] = row_load4(rs, (lane + k * 64) * 4);
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      v[k].x += a[k].x; v[k].y += a[k].y; v[k].z += a[k].z; v[k].w += a[k].w;
    }
  }
  if (bias) {
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      v[k].x += b4[k].x; v[k].y += b4[k].y; v[k].z += b4[k].z; v[k].w += b4[k].w;
    }
  }
  float mx = -INFINITY;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int c = (lane + k * 64) * 4;
    mx = fmaxf(mx, (c < cols) ? v[k].x : -INFINITY);
    mx = fmaxf(mx, (c + 1 < cols) ? v[k].y : -INFINITY);
    mx = fmaxf(mx, (c + 2 < cols) ? v[k].z : -INFINITY);
    mx = fmaxf(mx, (c + 3 < cols) ? v[k].w : -INFINITY);
  }
  mx = kh::wave_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int c = (lane + k * 64) * 4;
    v[k].x = (c < cols) ? expf(v[k].x - mx) : 0.f;
    v[k].y = (c + 1 < cols) ? expf(v[k].y - mx) : 0.f;
    v[k].z = (c + 2 < cols) ? expf(v[k].z - mx) : 0.f;
    v[k].w = (c + 3 < cols) ? expf(v[k].w - mx) : 0.f;
    sum += (v[k].x + v[k].y) + (v[k].z + v[k].w);
  }
  sum = kh::wave_sum(sum);
  return 1.0f / sum;
}
template <int NV>
__global__ __launch_bounds__(256) void softmax_wave_kernel(float *y, int ldy, const float *x, int ldx, int nslab, size_t slab_stride,
                                                           const float *bias, int rows, int cols) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  float4 v[NV];
  const float inv = softmax_row_wave<NV>(v, x + (size_t)r * ldx, nslab, slab_stride, bias, cols);
  const RowOut rsy = row_out(y + (size_t)r * ldy, cols);
#pragma unroll
  for (int k = 0; k < NV; ++k) row_store4(rsy, (lane + k * 64) * 4, v[k].x * inv, v[k].y * inv, v[k].z * inv, v[k].w * inv);
}

__global__ __launch_bounds__(256) void softmax_part_kernel(const float *X, const int32_t *ids, float *Y,
                                                           MatrixDim d) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= d.cols) return;
  for (int r = blockIdx.y; r < d.rows; r += gridDim.y) {
    const size_t base = (size_t)r * d.stride;
    Y[base + c] = expf(X[base + c] - X[base + ids[r]]);
  }
}

// ---- row argmax ----------------------------------------------------------------------------
struct MaxId {
  float v;
  int32_t i;
};
// larger value wins; equal values keep the lower column index (first maximum of the row)
__device__ __forceinline__ MaxId better(MaxId a, MaxId b) {
  if (b.v > a.v || (b.v == a.v && b.i >= 0 && (a.i < 0 || b.i < a.i))) return b;
  return a;
}
__device__ __forceinline__ MaxId wave_best(MaxId m) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    MaxId other;
    other.v = __shfl_xor(m.v, o, 64);
    other.i = __shfl_xor(m.i, o, 64);
    m = better(m, other);
  }
  return m;
}
// Row argmax of p[0..cols): start value -1e21 / id -1 and a strict '>' test, like the
// reference's CPU loop (cu-math-inl.h:339-348).  Result valid in every thread.
__device__ __forceinline__ MaxId block_row_argmax(const float *p, int cols, float *sv, int32_t *si) {
  MaxId m;
  m.v = -1e21f;
  m.i = -1;
  for (int c = threadIdx.x; c < cols; c += blockDim.x) {
    const float v = p[c];
    if (v > m.v) {  // ascending c per thread: the first maximum is kept
      m.v = v;
      m.i = c;
    }
  }
  m = wave_best(m);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) {
    sv[wid] = m.v;
    si[wid] = m.i;
  }
  __syncthreads();
  MaxId b;
  b.v = sv[0];
  b.i = si[0];
  for (int w = 1; w < nw; ++w) {
    MaxId o;
    o.v = sv[w];
    o.i = si[w];
    b = better(b, o);
  }
  return b;
}

__global__ __launch_bounds__(256) void find_row_max_id_kernel(const float *mat, float *vec_val, int32_t *vec_id,
                                                              MatrixDim d) {
  __shared__ float sv[16];
  __shared__ int32_t si[16];
  const int r = blockIdx.x;
  MaxId b = block_row_argmax(mat + (size_t)r * d.stride, d.cols, sv, si);
  if (threadIdx.x == 0) {
    vec_id[r] = b.i;
    if (vec_val) vec_val[r] = b.v;
  }
}

// one wave per row for short rows of a tall matrix (cols <= 256 * NV, 16-byte aligned rows): same result
template <int NV>
__global__ __launch_bounds__(256) void find_row_max_id_wave_kernel(const float *mat, float *vec_val, int32_t *vec_id, MatrixDim d) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= d.rows) return;
  const __amdgpu_buffer_rsrc_t rs = row_rsrc(mat + (size_t)r * d.stride, d.cols);
  float4 v[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) v[k] = row_load4(rs, (lane + k * 64) * 4);
  MaxId m;
  m.v = -1e21f;
  m.i = -1;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int c = (lane + k * 64) * 4;
    const float e[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (c + j < d.cols && e[j] > m.v) {  // ascending columns per lane: the first maximum is kept
        m.v = e[j];
        m.i = c + j;
      }
  }
  m = wave_best(m);
  if (lane == 0) {
    vec_id[r] = m.i;
    if (vec_val) vec_val[r] = m.v;
  }
}

__global__ void diff_xent_kernel(const int32_t *tgt, float *mat, float *log_post, MatrixDim d) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= d.rows) return;
  const int t = tgt[r];
  if (t < 0 || t >= d.cols) {  // label outside the output layer: nothing to subtract from
    log_post[r] = logf(1e-20f);
    return;
  }
  const size_t idx = (size_t)r * d.stride + t;
  float v = mat[idx];
  const float fl = (v < 1e-20f) ? 1e-20f : v;
  log_post[r] = logf(fl);
  mat[idx] = v - 1.0f;
}

// ---- fused Xent::EvalVec -------------------------------------------------------------------
// Per-row results travel from the workgroup that made them to the one that adds them up inside one launch, so
// they are stored write-through and loaded past the L1 (device-scope relaxed atomics = sc1 stores / loads): no
// release / acquire fence is needed then, and a fence here would write back or invalidate a whole L2 while every
// workgroup has just dirtied its row of the error matrix (the row kernel took 20 us at 256 x 9000 with fences).
__device__ __forceinline__ void row_result_store(float *log_post, int32_t *max_id, int r, float lp, int32_t id) {
  __hip_atomic_store(log_post + r, lp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(max_id + r, id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// pass 2: counters and loss in a fixed order, by one workgroup
__device__ __forceinline__ void xent_accum_block(const int32_t *tgt, const float *log_post, const int32_t *max_id,
                                                 long long *counters, double *loss, int rows) {
  // the first 256 threads of the workgroup do the work (callers have 256 or more)
  __shared__ double sd[256];
  __shared__ int sc[256];
  const int t = threadIdx.x;
  if (t < 256) {
    double acc = 0.0;
    int correct = 0;
    int r = t;
    // eight rows' loads in flight, added in the order of the plain loop below (a frame cache's worth of rows would
    // otherwise be 128 memory round trips one after another on the last workgroup)
    for (; r + 7 * 256 < rows; r += 8 * 256) {
      float lp[8];
      int32_t mi[8], tg[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        lp[u] = __hip_atomic_load(log_post + r + u * 256, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        mi[u] = __hip_atomic_load(max_id + r + u * 256, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        tg[u] = tgt[r + u * 256];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        acc += (double)lp[u];
        correct += (mi[u] == tg[u]) ? 1 : 0;
      }
    }
    for (; r < rows; r += 256) {
      acc += (double)__hip_atomic_load(log_post + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      correct += (__hip_atomic_load(max_id + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == tgt[r]) ? 1 : 0;
    }
    sd[t] = acc;
    sc[t] = correct;
  }
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (t < o) {
      sd[t] += sd[t + o];
      sc[t] += sc[t + o];
    }
    __syncthreads();
  }
  if (t == 0) {
    counters[0] += sc[0];
    counters[1] += rows;
    loss[0] -= sd[0];
  }
}
// Called by every workgroup of a one-row-per-workgroup kernel after thread 0 stored the row's results with
// row_result_store: the workgroup that arrives last (a device-scope counter, back at zero afterwards) runs
// pass 2, so no second kernel sits on the stream between the loss and the first backward GEMM.  Thread 0 waits for
// its own write-through stores before it adds to the counter; the last arriver's other threads start loading
// behind the workgroup barrier that follows thread 0's add.  The sums are taken in the same order whichever
// workgroup that is.
struct XentTail {
  unsigned *ticket;
  long long *counters;
  double *loss;
};
__device__ __forceinline__ void xent_finish(const XentTail &x, const int32_t *tgt, const float *log_post,
                                            const int32_t *max_id, int rows) {
  __shared__ unsigned arrived;
  if (threadIdx.x == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this row's results have reached the L2
    arrived = __hip_atomic_fetch_add(x.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (arrived != gridDim.x - 1) return;
  xent_accum_block(tgt, log_post, max_id, x.counters, x.loss, rows);
  if (threadIdx.x == 0) __hip_atomic_store(x.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// pass 1: one block per row: argmax, log posterior of the target, diff = post - onehot
__global__ __launch_bounds__(256) void xent_rows_kernel(const float *post, const int32_t *tgt, float *diff,
                                                        float *log_post, int32_t *max_id, MatrixDim d, XentTail tail) {
  __shared__ float sv[16];
  __shared__ int32_t si[16];
  const int r = blockIdx.x;
  const float *p = post + (size_t)r * d.stride;
  float *q = diff + (size_t)r * d.stride;
  const int t = tgt[r];
  const bool t_ok = (t >= 0 && t < d.cols);  // a label outside the output layer counts as posterior 0
  // argmax on the unmodified posteriors (diff may alias post, so finish reading first)
  MaxId b = block_row_argmax(p, d.cols, sv, si);
  const float pt = t_ok ? p[t] : 0.0f;
  __syncthreads();
  if (q != p) {
    for (int c = threadIdx.x; c < d.cols; c += 256) q[c] = (c == t) ? p[c] - 1.0f : p[c];
  } else if (threadIdx.x == 0 && t_ok) {
    q[t] = pt - 1.0f;
  }
  if (threadIdx.x == 0) row_result_store(log_post, max_id, r, logf(pt < 1e-20f ? 1e-20f : pt), b.i);
  xent_finish(tail, tgt, log_post, max_id, d.rows);
}
// pass 1, fast form: the row is read once with 16-byte loads into registers (NV float4 per thread,
// all loads in flight together), then argmax and diff come out of the registers.  Needs 16-byte
// aligned rows, diff not aliasing post and cols <= 1024 * NV.
template <int NV>
__global__ __launch_bounds__(256) void xent_rows_vec_kernel(const float *post, const int32_t *tgt, float *diff,
                                                            float *log_post, int32_t *max_id, MatrixDim d,
                                                            XentTail tail) {
  __shared__ float sv[16];
  __shared__ int32_t si[16];
  const int r = blockIdx.x;
  const float *p = post + (size_t)r * d.stride;
  float *q = diff + (size_t)r * d.stride;
  const int t = tgt[r];
  const bool t_ok = (t >= 0 && t < d.cols);
  float4 v[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int c = (threadIdx.x + k * 256) * 4;
    // whole float4s inside the padded row are readable; columns >= cols are masked below
    v[k] = (c < d.stride) ? *reinterpret_cast<const float4 *>(p + c) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  MaxId m;
  m.v = -1e21f;
  m.i = -1;
  float pt = 0.0f;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int c = (threadIdx.x + k * 256) * 4;
    float e[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (c + j < d.cols) {
        if (e[j] > m.v) {
          m.v = e[j];
          m.i = c + j;
        }
        if (c + j == t) {
          pt = e[j];
          e[j] -= 1.0f;
        }
      }
    }
    if (c + 3 < d.cols) {
      *reinterpret_cast<float4 *>(q + c) = make_float4(e[0], e[1], e[2], e[3]);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (c + j < d.cols) q[c + j] = e[j];
    }
  }
  // per-thread columns ascend with k, so `m` holds the thread's first maximum; combine as in
  // block_row_argmax (lower index wins ties)
  m = wave_best(m);
  // the target's posterior lives in exactly one thread
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) pt += __shfl_xor(pt, o, 64);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  __shared__ float spt[4];
  if (lane == 0) {
    sv[wid] = m.v;
    si[wid] = m.i;
    spt[wid] = pt;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    MaxId b;
    b.v = sv[0];
    b.i = si[0];
    float ptot = spt[0];
    for (int w = 1; w < 4; ++w) {
      MaxId o;
      o.v = sv[w];
      o.i = si[w];
      b = better(b, o);
      ptot += spt[w];
    }
    const float ptv = t_ok ? ptot : 0.0f;
    row_result_store(log_post, max_id, r, logf(ptv < 1e-20f ? 1e-20f : ptv), b.i);
  }
  xent_finish(tail, tgt, log_post, max_id, d.rows);
}
// pass 1 for short rows of a tall matrix (cols <= 256 * NV): one WAVE per row, four rows and ONE arrival per
// workgroup.  Arg-max, target posterior and diff are what xent_rows_vec_kernel computes, and the totals are added up
// by the same xent_accum_block: the same bits.
template <int NV>
__global__ __launch_bounds__(256) void xent_rows_wave_kernel(const float *post, const int32_t *tgt, float *diff, float *log_post,
                                                             int32_t *max_id, MatrixDim d, int rows_per_wave, XentTail tail) {
  const int lane = threadIdx.x & 63;
  // (a wave takes rows_per_wave rows one after another: an arrival per row, or per four, is a queue on ONE word --
  // about 90 adds per microsecond -- that 8192 workgroups of a frame cache spend longer in than in their rows)
  for (int it = 0; it < rows_per_wave; ++it) {
    const int r = (blockIdx.x * rows_per_wave + it) * 4 + (threadIdx.x >> 6);
    if (r >= d.rows) break;
    const int t = tgt[r];
    const bool t_ok = (t >= 0 && t < d.cols);
    const __amdgpu_buffer_rsrc_t rs = row_rsrc(post + (size_t)r * d.stride, d.cols);
    const RowOut rq = row_out(diff + (size_t)r * d.stride, d.cols);
    float4 v[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = row_load4(rs, (lane + k * 64) * 4);
    MaxId m;
    m.v = -1e21f;
    m.i = -1;
    float pt = 0.0f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int c = (lane + k * 64) * 4;
      float e[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (c + j < d.cols) {
          if (e[j] > m.v) {
            m.v = e[j];
            m.i = c + j;
          }
          if (c + j == t) {
            pt = e[j];
            e[j] -= 1.0f;
          }
        } else {
          e[j] = 0.f;  // (row padding inside the last group of 4)
        }
      }
      row_store4(rq, c, e[0], e[1], e[2], e[3]);
    }
    m = wave_best(m);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) pt += __shfl_xor(pt, o, 64);  // the target's posterior lives in exactly one lane
    if (lane == 0) {
      const float ptv = t_ok ? pt : 0.0f;
      row_result_store(log_post, max_id, r, logf(ptv < 1e-20f ? 1e-20f : ptv), m.i);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's row results have reached the L2
  // one arrival for the workgroup's rows, behind the waits of all its waves
  __shared__ unsigned arrived;
  __syncthreads();
  if (threadIdx.x == 0) arrived = __hip_atomic_fetch_add(tail.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (arrived != gridDim.x - 1) return;
  xent_accum_block(tgt, log_post, max_id, tail.counters, tail.loss, d.rows);
  if (threadIdx.x == 0) __hip_atomic_store(tail.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Softmax finish of the output layer's GEMM and Xent::EvalVec in one pass: the posteriors of a row
// never leave the registers between the two (they are stored only when the caller wants them).
// Arg-max, target posterior and diff are taken from the same rounded posteriors the two-kernel
// path stores and re-reads, so counts and sums are identical to it.
struct XentFuse {
  const int32_t *tgt;
  float *diff;
  int lddiff;
  float *log_post;
  int32_t *max_id;
  int write_post;
  int rows;
  XentTail tail;
};
template <int NV>
__global__ __launch_bounds__(kSoftmaxThreads) void softmax_xent_vec_kernel(float *y, int ldy, const float *x, int ldx,
                                                                           int nslab, size_t slab_stride,
                                                                           const float *bias, int cols, XentFuse xf) {
  __shared__ float red[16];
  __shared__ float sv[kSoftmaxThreads / 64];
  __shared__ int32_t si[kSoftmaxThreads / 64];
  __shared__ float spt[kSoftmaxThreads / 64];
  const int r = blockIdx.x;
  float4 v[NV];
  const float inv = softmax_row_regs<NV>(v, x + (size_t)r * ldx, nslab, slab_stride, bias, cols, red);
  const int t = xf.tgt[r];
  const bool t_ok = (t >= 0 && t < cols);
  const RowOut rsy = row_out(y + (size_t)r * ldy, xf.write_post ? cols : 0);  // 0 columns: nothing is written
  const RowOut rsd = row_out(xf.diff + (size_t)r * xf.lddiff, cols);
  // Order of the rest: arg-max and target posterior first (registers only), the row's results and this
  // workgroup's arrival next, the row stores after that -- so the write acknowledgement and the counter's round
  // trip (two trips to the memory side) pass while the 2 x 36 KB of the row are being stored, not after them --
  // and last the check whether this workgroup arrived last.  (Stores first, the barrier behind the arg-max waited
  // for every store of every thread, and the two round trips came on top.)
  MaxId m;
  m.v = -1e21f;
  m.i = -1;
  float pt = 0.0f;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    // posterior = v * inv rounded, then - 1 at the target: two roundings, as when the posterior is read back
    // from memory by the two-kernel form (a fused multiply-add here would differ in the last bit)
#pragma clang fp contract(off)
    const int c = (threadIdx.x + k * kSoftmaxThreads) * 4;
    const float e[4] = {v[k].x * inv, v[k].y * inv, v[k].z * inv, v[k].w * inv};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool in = c + j < cols;
      const bool up = in && e[j] > m.v;  // ascending columns, strict: the first maximum stays
      m.v = up ? e[j] : m.v;
      m.i = up ? c + j : m.i;
      pt = (in && (c + j == t)) ? e[j] : pt;
    }
  }
  m = wave_best(m);  // per-thread columns ascend, lower index wins ties: the first maximum of the row
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) pt += __shfl_xor(pt, o, 64);  // the target's posterior lives in one thread
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) {
    sv[wid] = m.v;
    si[wid] = m.i;
    spt[wid] = pt;
  }
  __syncthreads();
  __shared__ unsigned arrived;
  unsigned arrival = 0;
  if (threadIdx.x == 0) {
    MaxId b;
    b.v = sv[0];
    b.i = si[0];
    float ptot = spt[0];
    for (int w = 1; w < kSoftmaxThreads / 64; ++w) {
      MaxId o;
      o.v = sv[w];
      o.i = si[w];
      b = better(b, o);
      ptot += spt[w];
    }
    const float ptv = t_ok ? ptot : 0.0f;
    row_result_store(xf.log_post, xf.max_id, r, logf(ptv < 1e-20f ? 1e-20f : ptv), b.i);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this row's results have reached the L2
    arrival = __hip_atomic_fetch_add(xf.tail.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
#pragma unroll
  for (int k = 0; k < NV; ++k) {
#pragma clang fp contract(off)
    const int c = (threadIdx.x + k * kSoftmaxThreads) * 4;
    float e[4] = {v[k].x * inv, v[k].y * inv, v[k].z * inv, v[k].w * inv};
    row_store4(rsy, c, e[0], e[1], e[2], e[3]);
#pragma unroll
    for (int j = 0; j < 4; ++j) e[j] = (c + j < cols && c + j == t) ? e[j] - 1.0f : e[j];
    row_store4(rsd, c, e[0], e[1], e[2], e[3]);
  }
  asm volatile("" ::: "memory");
  if (threadIdx.x == 0) arrived = arrival;
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // (LDS only: the row stores stay in flight)
  if (arrived != gridDim.x - 1) return;
  xent_accum_block(xf.tgt, xf.log_post, xf.max_id, xf.tail.counters, xf.tail.loss, xf.rows);
  if (threadIdx.x == 0) __hip_atomic_store(xf.tail.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The same for rows of up to 2048 columns (an RM-sized output layer): one wave per row, four rows and ONE arrival per
// workgroup, no block reduction.  Same order as above -- results and arrival first, row stores behind them -- with
// the wait for the results counted past the row stores (the memory counter retires in order).  Posteriors, arg-max,
// target posterior and diff have the bits of softmax_wave_kernel followed by the loss kernels.
template <int NV>
__global__ __launch_bounds__(256) void softmax_xent_wave_kernel(float *y, int ldy, const float *x, int ldx, int nslab,
                                                                size_t slab_stride, const float *bias, int cols, XentFuse xf) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r < xf.rows) {
    float4 v[NV];
    const float inv = softmax_row_wave<NV>(v, x + (size_t)r * ldx, nslab, slab_stride, bias, cols);
    const int t = xf.tgt[r];
    const bool t_ok = (t >= 0 && t < cols);
    const RowOut rsy = row_out(y + (size_t)r * ldy, xf.write_post ? cols : 0);  // 0 columns: nothing is written
    const RowOut rsd = row_out(xf.diff + (size_t)r * xf.lddiff, cols);
    MaxId m;
    m.v = -1e21f;
    m.i = -1;
    float pt = 0.0f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
#pragma clang fp contract(off)
      const int c = (lane + k * 64) * 4;
      const float e[4] = {v[k].x * inv, v[k].y * inv, v[k].z * inv, v[k].w * inv};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool in = c + j < cols;
        const bool up = in && e[j] > m.v;  // ascending columns, strict: the first maximum stays
        m.v = up ? e[j] : m.v;
        m.i = up ? c + j : m.i;
        pt = (in && (c + j == t)) ? e[j] : pt;
      }
    }
    m = wave_best(m);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) pt += __shfl_xor(pt, o, 64);  // the target's posterior lives in one lane
    if (lane == 0) {
      const float ptv = t_ok ? pt : 0.0f;
      row_result_store(xf.log_post, xf.max_id, r, logf(ptv < 1e-20f ? 1e-20f : ptv), m.i);
    }
#pragma unroll
    for (int k = 0; k < NV; ++k) {
#pragma clang fp contract(off)
      const int c = (lane + k * 64) * 4;
      float e[4] = {v[k].x * inv, v[k].y * inv, v[k].z * inv, v[k].w * inv};
      row_store4(rsy, c, e[0], e[1], e[2], e[3]);
#pragma unroll
      for (int j = 0; j < 4; ++j) e[j] = (c + j < cols && c + j == t) ? e[j] - 1.0f : e[j];
      row_store4(rsd, c, e[0], e[1], e[2], e[3]);
    }
    // the two result stores have reached the L2 once all but the 2 * NV row stores behind them have retired
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NV) : "memory");
  }
  __shared__ unsigned arrived;
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // (the row stores stay in flight)
  if (threadIdx.x == 0) arrived = __hip_atomic_fetch_add(xf.tail.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  if (arrived != gridDim.x - 1) return;
  xent_accum_block(xf.tgt, xf.log_post, xf.max_id, xf.tail.counters, xf.tail.loss, xf.rows);
  if (threadIdx.x == 0) __hip_atomic_store(xf.tail.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- fused Mse / MseProgress Eval ------------------------------------------------------------
// One workgroup per row, one launch: kh::mse_row_and_total (a second one-workgroup kernel for the total cost
// ~2.5-4 us per call on the critical path of an RBM step).
__global__ __launch_bounds__(256) void mse_rows_kernel(const float *a, const float *b, float *diff, float *row_sq,
                                                       MatrixDim d, unsigned *ticket, double *loss) {
  __shared__ float red[16];
  __shared__ double sd[256];
  __shared__ unsigned arrived;
  const int r = blockIdx.x;
  const size_t base = (size_t)r * d.stride;
  kh::mse_row_and_total(a + base, b + base, diff ? diff + base : nullptr, d.cols, row_sq, r, d.rows, ticket, loss, red,
                        sd, &arrived);
}

}  // namespace

namespace kh {
int softmax_slabs_launch(hipStream_t st, float *y, int ldy, const float *x, int ldx, int nslab, size_t slab_stride,
                         const float *bias, int rows, int cols, const XentArgs *xent) {
  const bool vec = (ldx % 4 == 0) && (ldy % 4 == 0) && (slab_stride % 4 == 0) && kh::aligned16(x) && kh::aligned16(y) &&
                   (!bias || kh::aligned16(bias));
  if (xent && vec && cols <= 4 * kSoftmaxThreads * 5 && xent->lddiff % 4 == 0 && kh::aligned16(xent->diff) &&
      xent->diff != y) {
    XentFuse xf;
    xf.tgt = xent->tgt;
    xf.diff = xent->diff;
    xf.lddiff = xent->lddiff;
    xf.log_post = xent->log_post;
    xf.max_id = xent->max_id;
    xf.write_post = xent->write_post;
    xf.rows = rows;
    xf.tail.counters = xent->counters;
    xf.tail.loss = xent->loss;
    int rc = kh::stream_ticket(st, &xf.tail.ticket);
    if (rc != KH_OK) return rc;
    if (!xf.log_post || !xf.max_id) {
      // per-row results nobody asked for go behind the slabs in the stream's workspace
      void *ws;
      const size_t slabs = (size_t)nslab * slab_stride * sizeof(float);
      rc = kh::ensure_workspace(st, slabs + (size_t)rows * 8, &ws);
      if (rc != KH_OK) return rc;
      if ((const void *)x != ws) return KH_ERR_INVALID_ARG;  // the slabs must be that workspace's (they are: run_gemm)
      if (!xf.log_post) xf.log_post = (float *)((char *)ws + slabs);
      if (!xf.max_id) xf.max_id = (int32_t *)((char *)ws + slabs + (size_t)rows * 4);
    }
    if (cols <= 2048) {
      // short rows: a wave per row (the same choice as the stand-alone softmax below: same posteriors)
      const dim3 grid((rows + 3) / 4), block(256);
      if (cols <= 512) hipLaunchKernelGGL((softmax_xent_wave_kernel<2>), grid, block, 0, st, y, ldy, x, ldx, nslab, slab_stride, bias, cols, xf);
      else if (cols <= 1024) hipLaunchKernelGGL((softmax_xent_wave_kernel<4>), grid, block, 0, st, y, ldy, x, ldx, nslab, slab_stride, bias, cols, xf);
      else if (cols <= 1536) hipLaunchKernelGGL((softmax_xent_wave_kernel<6>), grid, block, 0, st, y, ldy, x, ldx, nslab, slab_stride, bias, cols, xf);
      else hipLaunchKernelGGL((softmax_xent_wave_kernel<8>), grid, block, 0, st, y, ldy, x, ldx, nslab, slab_stride, bias, cols, xf);
    } else if (cols <= 4 * kSoftmaxThreads * 2)
      hipLaunchKernelGGL((softmax_xent_vec_kernel<2>), dim3(rows), dim3(kSoftmaxThreads), 0, st, y, ldy, x, ldx, nslab,
                         slab_stride, bias, cols, xf);
    else if (cols <= 4 * kSoftmaxThreads * 3)
      hipLaunchKernelGGL((softmax_xent_vec_kernel<3>), dim3(rows), dim3(kSoftmaxThreads), 0, st, y, ldy, x, ldx, nslab,
                         slab_stride, bias, cols, xf);
    else
      hipLaunchKernelGGL((softmax_xent_vec_kernel<5>), dim3(rows), dim3(kSoftmaxThreads), 0, st, y, ldy, x, ldx, nslab,
                         slab_stride, bias, cols, xf);
    return kh::launch_status();
  }
  // Short rows: one wave per row.  Chosen by the row LENGTH only, so that a row's bits do not depend on how many
  // rows share the call.
  if (vec && cols <= 2048) {
    const dim3 grid((rows + 3) / 4), block(256);
    if (cols <= 512) hipLaunchKernelGGL((softmax_wave_kernel<2>), grid, block, 0, st, y, ldy, x, ldx, nslab, slab_stride, bias, rows, cols);
    else if (cols <= 1024) hipLaunchKernelGGL((softmax_wave_kernel<4>), grid, block, 0, st, y, ldy, x, ldx, nslab, slab_stride, bias, rows, cols);
    else if (cols <= 1536) hipLaunchKernelGGL((softmax_wave_kernel<6>), grid, block, 0, st, y, ldy, x, ldx, nslab, slab_stride, bias, rows, cols);
    else hipLaunchKernelGGL((softmax_wave_kernel<8>), grid, block, 0, st, y, ldy, x, ldx, nslab, slab_stride, bias, rows, cols);
    int rc = kh::launch_status();
    if (rc != KH_OK || !xent) return rc;
    // (a fused request whose shape the fused kernel did not take: the loss as a second kernel)
    MatrixDim d = {rows, cols, ldy};
    if (xent->lddiff != ldy) return KH_ERR_INVALID_ARG;
    return hipF_xent_eval((khStream_t)st, y, xent->tgt, xent->diff, xent->log_post, xent->max_id, xent->counters,
                          xent->loss, d);
  }
  if (vec && cols <= 4 * kSoftmaxThreads * 2)
    hipLaunchKernelGGL((softmax_vec_kernel<2>), dim3(rows), dim3(kSoftmaxThreads), 0, st, y, ldy, x, ldx, nslab,
                       slab_stride, bias, cols);
  else if (vec && cols <= 4 * kSoftmaxThreads * 5)
    hipLaunchKernelGGL((softmax_vec_kernel<5>), dim3(rows), dim3(kSoftmaxThreads), 0, st, y, ldy, x, ldx, nslab,
                       slab_stride, bias, cols);
  else if (cols <= kSoftmaxLdsFloats)
    hipLaunchKernelGGL((softmax_kernel<true>), dim3(rows), dim3(kSoftmaxThreads), 0, st, y, ldy, x, ldx, nslab,
                       slab_stride, bias, cols);
  else
    hipLaunchKernelGGL((softmax_kernel<false>), dim3(rows), dim3(kSoftmaxThreads), 0, st, y, ldy, x, ldx, nslab,
                       slab_stride, bias, cols);
  int rc = kh::launch_status();
  if (rc != KH_OK || !xent) return rc;
  // two-kernel form of the same thing (shapes the fused kernel does not take)
  MatrixDim d = {rows, cols, ldy};
  if (xent->lddiff != ldy) return KH_ERR_INVALID_ARG;
  return hipF_xent_eval((khStream_t)st, y, xent->tgt, xent->diff, xent->log_post, xent->max_id, xent->counters,
                        xent->loss, d);
}
}  // namespace kh

KH_EXPORT int hipF_add_row_sum_mat(khStream_t s, float alpha, const float *mat, MatrixDim d, float beta,
                                   float *vec) {
  KH_CHECK_INIT();
  KH_REQUIRE(kh::dim_ok(d));
  if (d.cols == 0) return KH_OK;
  KH_REQUIRE(vec != nullptr && (d.rows == 0 || mat != nullptr));
  hipStream_t st = kh::pick(s);
  // Up to a few bunches of rows: one workgroup per 64 columns (the sums of a 256-frame bunch keep their bits).
  // Taller: two stages over all compute units (col_sums_slab_kernel).
  if (d.rows < 2048) {
    hipLaunchKernelGGL(col_sums_kernel, dim3((d.cols + 63) / 64), dim3(1024), 0, st, alpha, mat, d, beta, vec);
    return kh::launch_status();
  }
  const bool vec4 = d.stride % 4 == 0 && kh::aligned16(mat);
  const int nv = !vec4 ? 0 : d.cols <= 256 ? 1 : d.cols <= 512 ? 2 : d.cols <= 1024 ? 4 : d.cols <= 1536 ? 6 : 8;
  const int gx = vec4 ? (d.cols + 256 * nv - 1) / (256 * nv) : (d.cols + 63) / 64;
  int nslab = 256 / gx;  // about one workgroup of 16 waves per compute unit; stage 2 then sums at most 256 rows
  if (nslab > d.rows / 64) nslab = d.rows / 64;
  if (nslab < 1) nslab = 1;
  const int rows_per_slab = (d.rows + nslab - 1) / nslab;
  nslab = (d.rows + rows_per_slab - 1) / rows_per_slab;
  const int ldpart = (d.cols + 3) & ~3;
  void *ws;
  int rc = kh::ensure_workspace(st, (size_t)nslab * ldpart * sizeof(float), &ws);
  if (rc != KH_OK) return rc;
  const dim3 grid(gx, nslab), block(nv ? 1024 : 256);
  float *part = (float *)ws;
  switch (nv) {
    case 0: hipLaunchKernelGGL(col_sums_slab_scalar_kernel, grid, block, 0, st, mat, d, rows_per_slab, part, ldpart); break;
    case 1: hipLaunchKernelGGL(col_sums_slab_kernel<1>, grid, block, 0, st, mat, d, rows_per_slab, part, ldpart); break;
    case 2: hipLaunchKernelGGL(col_sums_slab_kernel<2>, grid, block, 0, st, mat, d, rows_per_slab, part, ldpart); break;
    case 4: hipLaunchKernelGGL(col_sums_slab_kernel<4>, grid, block, 0, st, mat, d, rows_per_slab, part, ldpart); break;
    case 6: hipLaunchKernelGGL(col_sums_slab_kernel<6>, grid, block, 0, st, mat, d, rows_per_slab, part, ldpart); break;
    default: hipLaunchKernelGGL(col_sums_slab_kernel<8>, grid, block, 0, st, mat, d, rows_per_slab, part, ldpart); break;
  }
  MatrixDim dp = {nslab, d.cols, ldpart};
  hipLaunchKernelGGL(col_sums_kernel, dim3((d.cols + 63) / 64), dim3(1024), 0, st, alpha, (const float *)part, dp, beta, vec);
  return kh::launch_status();
}

KH_EXPORT int hipF_add_col_sum_mat(khStream_t s, float alpha, const float *mat, MatrixDim d, float beta,
                                   float *vec) {
  KH_CHECK_INIT();
  KH_REQUIRE(kh::dim_ok(d));
  if (d.rows == 0) return KH_OK;
  KH_REQUIRE(vec != nullptr && (d.cols == 0 || mat != nullptr));
  hipLaunchKernelGGL(row_sums_kernel, dim3(d.rows), dim3(256), 0, kh::pick(s), alpha, mat, d, beta, vec);
  return kh::launch_status();
}

KH_EXPORT int hipF_softmax(khStream_t s, float *y, const float *x, MatrixDim d) {
  KH_CHECK_INIT();
  KH_REQUIRE(kh::dim_ok(d));
  if (d.rows == 0 || d.cols == 0) return KH_OK;
  KH_REQUIRE(y != nullptr && x != nullptr);
  return kh::softmax_slabs_launch(kh::pick(s), y, d.stride, x, d.stride, 1, 0, nullptr, d.rows, d.cols, nullptr);
}

KH_EXPORT int hipF_softmax_part(khStream_t s, const float *X, const int32_t *vec_ids, float *Y, MatrixDim d) {
  KH_CHECK_INIT();
  KH_REQUIRE(kh::dim_ok(d));
  if (d.rows == 0 || d.cols == 0) return KH_OK;
  KH_REQUIRE(X && vec_ids && Y);
  int gx = (d.cols + 255) / 256, gy = d.rows;
  int cap = 8192 / gx;
  if (gy > cap) gy = cap < 1 ? 1 : cap;
  hipLaunchKernelGGL(softmax_part_kernel, dim3(gx, gy), dim3(256), 0, kh::pick(s), X, vec_ids, Y, d);
  return kh::launch_status();
}

KH_EXPORT int hipF_find_row_max_id(khStream_t s, const float *mat, float *vec_val, int32_t *vec_id, MatrixDim d) {
  KH_CHECK_INIT();
  KH_REQUIRE(kh::dim_ok(d));
  if (d.rows == 0) return KH_OK;
  KH_REQUIRE(vec_id != nullptr && (d.cols == 0 || mat != nullptr));
  hipStream_t st = kh::pick(s);
  if (d.rows >= 1024 && d.cols > 0 && d.cols <= 2048 && d.stride % 4 == 0 && kh::aligned16(mat)) {
    const dim3 grid((d.rows + 3) / 4), block(256);
    if (d.cols <= 512) hipLaunchKernelGGL(find_row_max_id_wave_kernel<2>, grid, block, 0, st, mat, vec_val, vec_id, d);
    else if (d.cols <= 1024) hipLaunchKernelGGL(find_row_max_id_wave_kernel<4>, grid, block, 0, st, mat, vec_val, vec_id, d);
    else if (d.cols <= 1536) hipLaunchKernelGGL(find_row_max_id_wave_kernel<6>, grid, block, 0, st, mat, vec_val, vec_id, d);
    else hipLaunchKernelGGL(find_row_max_id_wave_kernel<8>, grid, block, 0, st, mat, vec_val, vec_id, d);
    return kh::launch_status();
  }
  hipLaunchKernelGGL(find_row_max_id_kernel, dim3(d.rows), dim3(256), 0, st, mat, vec_val, vec_id, d);
  return kh::launch_status();
}

KH_EXPORT int hipF_diff_xent(khStream_t s, const int32_t *vec_tgt, float *mat_net_out, float *vec_log_post,
                             MatrixDim d) {
  KH_CHECK_INIT();
  KH_REQUIRE(kh::dim_ok(d));
  if (d.rows == 0) return KH_OK;
  KH_REQUIRE(vec_tgt && mat_net_out && vec_log_post && d.cols > 0);
  hipLaunchKernelGGL(diff_xent_kernel, dim3((d.rows + 127) / 128), dim3(128), 0, kh::pick(s), vec_tgt, mat_net_out,
                     vec_log_post, d);
  return kh::launch_status();
}

namespace {
// pass 1 + "last workgroup sums up" of Xent::EvalVec; lp / mi / ticket are valid device addresses
int xent_eval_launch(hipStream_t st, const float *post, const int32_t *tgt, float *diff, float *lp, int32_t *mi,
                     unsigned *ticket, long long *counters, double *loss, MatrixDim d) {
  XentTail tail;
  tail.counters = counters;
  tail.loss = loss;
  tail.ticket = ticket;
  const bool vec = (post != diff) && (d.stride % 4 == 0) && kh::aligned16(post) && kh::aligned16(diff);
  if (vec && d.rows >= 1024 && d.cols <= 2048) {
    // many short rows: a wave per row, one arrival per four rows
    int rpw = d.rows / 1024;  // a few hundred to a thousand workgroups (arrivals), 1..8 rows per wave
    rpw = rpw < 1 ? 1 : rpw > 8 ? 8 : rpw;
    const dim3 grid((d.rows + 4 * rpw - 1) / (4 * rpw)), block(256);
    if (d.cols <= 512) hipLaunchKernelGGL(xent_rows_wave_kernel<2>, grid, block, 0, st, post, tgt, diff, lp, mi, d, rpw, tail);
    else if (d.cols <= 1024) hipLaunchKernelGGL(xent_rows_wave_kernel<4>, grid, block, 0, st, post, tgt, diff, lp, mi, d, rpw, tail);
    else if (d.cols <= 1536) hipLaunchKernelGGL(xent_rows_wave_kernel<6>, grid, block, 0, st, post, tgt, diff, lp, mi, d, rpw, tail);
    else hipLaunchKernelGGL(xent_rows_wave_kernel<8>, grid, block, 0, st, post, tgt, diff, lp, mi, d, rpw, tail);
  } else if (vec && d.cols <= 4096)
    hipLaunchKernelGGL(xent_rows_vec_kernel<4>, dim3(d.rows), dim3(256), 0, st, post, tgt, diff, lp, mi, d, tail);
  else if (vec && d.cols <= 10240)
    hipLaunchKernelGGL(xent_rows_vec_kernel<10>, dim3(d.rows), dim3(256), 0, st, post, tgt, diff, lp, mi, d, tail);
  else if (vec && d.cols <= 16384)
    hipLaunchKernelGGL(xent_rows_vec_kernel<16>, dim3(d.rows), dim3(256), 0, st, post, tgt, diff, lp, mi, d, tail);
  else
    hipLaunchKernelGGL(xent_rows_kernel, dim3(d.rows), dim3(256), 0, st, post, tgt, diff, lp, mi, d, tail);
  return kh::launch_status();
}
}  // namespace

KH_EXPORT int hipF_xent_eval(khStream_t s, const float *post, const int32_t *tgt, float *diff, float *log_post,
                             int32_t *max_id, long long *counters, double *loss, MatrixDim d) {
  KH_CHECK_INIT();
  KH_REQUIRE(kh::dim_ok(d));
  if (d.rows == 0) return KH_OK;
  KH_REQUIRE(post && tgt && diff && counters && loss && d.cols > 0);
  // per-row results go to the caller's vectors or to library scratch
  float *lp = log_post;
  int32_t *mi = max_id;
  if (!lp || !mi) {
    void *ws;
    int st = kh::ensure_workspace(kh::pick(s), (size_t)d.rows * 8, &ws);
    if (st != KH_OK) return st;
    if (!lp) lp = (float *)ws;
    if (!mi) mi = (int32_t *)((char *)ws + (size_t)d.rows * 4);
  }
  hipStream_t st = kh::pick(s);
  unsigned *ticket;
  {
    int rc = kh::stream_ticket(st, &ticket);
    if (rc != KH_OK) return rc;
  }
  return xent_eval_launch(st, post, tgt, diff, lp, mi, ticket, counters, loss, d);
}

// hipF_xent_eval for callers that cannot allocate: same kernels, the arrival counter and the per-row
// scratch live in the file-scope `scratch` array.  No hipMalloc / hipFree, no implicit hipK_init.
KH_EXPORT int hipF_xent_eval_static(khStream_t s, const float *post, const int32_t *tgt, float *diff,
                                    float *log_post, int32_t *max_id, long long *counters, double *loss,
                                    MatrixDim d) {
  if (!kh::ctx().initialised) return KH_ERR_NOT_INITIALISED;  // (hipK_init allocates; the caller runs it beforehand)
  KH_REQUIRE(kh::dim_ok(d));
  if (d.rows == 0) return KH_OK;
  KH_REQUIRE(post && tgt && diff && counters && loss && d.cols > 0);
  if (d.rows > kStaticXentRows) {
    snprintf(kh::ctx().last_msg, sizeof(kh::ctx().last_msg),
             "hipF_xent_eval_static: %d rows, the static scratch holds at most %d", d.rows, kStaticXentRows);
    return KH_ERR_TOO_LARGE;
  }
  const int dev = kh::ctx().device;
  unsigned char *ptr = (dev >= 0 && dev < kMaxDevices) ? g_scratch_ptr[dev] : nullptr;
  if (!ptr) return KH_ERR_NOT_INITIALISED;
  hipStream_t st = kh::pick(s);
  // the counter must start at zero whatever an earlier (possibly aborted) call left behind
  KH_HIP(hipMemsetAsync(ptr, 0, kStaticTicketBytes, st));
  float *lp = log_post ? log_post : (float *)(ptr + kStaticTicketBytes);
  int32_t *mi = max_id ? max_id : (int32_t *)(ptr + kStaticTicketBytes + (size_t)kStaticXentRows * 4);
  return xent_eval_launch(st, post, tgt, diff, lp, mi, (unsigned *)ptr, counters, loss, d);
}

KH_EXPORT int hipF_mse_eval(khStream_t s, const float *a, const float *b, float *diff, float *row_sq, double *loss,
                            MatrixDim d) {
  KH_CHECK_INIT();
  KH_REQUIRE(kh::dim_ok(d));
  if (d.rows == 0) return KH_OK;
  KH_REQUIRE(a && b && loss);
  float *rs = row_sq;
  if (!rs) {
    void *ws;
    int st = kh::ensure_workspace(kh::pick(s), (size_t)d.rows * 4, &ws);
    if (st != KH_OK) return st;
    rs = (float *)ws;
  }
  hipStream_t st = kh::pick(s);
  unsigned *ticket;
  {
    int rc = kh::stream_ticket(st, &ticket);
    if (rc != KH_OK) return rc;
  }
  hipLaunchKernelGGL(mse_rows_kernel, dim3(d.rows), dim3(256), 0, st, a, b, diff, rs, d, ticket, loss);
  return kh::launch_status();
}
