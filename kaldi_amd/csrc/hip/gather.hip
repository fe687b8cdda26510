// Row / column gathers: frame shuffle (cu::Randomize), splice (cu::Expand), column select
// (cu::Copy).  src/cudamatrix/cu-kernels.cu:627-669.  HBM-bound copies: each output row is
// written as one contiguous run, 16 bytes per lane where the layout allows.
#include "common.h"

namespace {

// y[r, :] = x[copy_from[r], :]; one block per output row
template <int VEC>
__global__ __launch_bounds__(256) void randomize_kernel(float *y, const float *x, const int32_t *copy_from,
                                                        MatrixDim d_out, MatrixDim d_in) {
  for (int r = blockIdx.x; r < d_out.rows; r += gridDim.x) {
    const int src = copy_from[r];
    if (src < 0 || src >= d_in.rows) continue;  // never read outside x
    const float *px = x + (size_t)src * d_in.stride;
    float *py = y + (size_t)r * d_out.stride;
    if (VEC == 4) {
      for (int c = threadIdx.x * 4; c < d_out.cols; c += 1024)
        *reinterpret_cast<float4 *>(py + c) = *reinterpret_cast<const float4 *>(px + c);
    } else {
      for (int c = threadIdx.x; c < d_out.cols; c += 256) py[c] = px[c];
    }
  }
}

// The same for a whole cache of frames: the 16-byte elements of all rows are dealt out to the threads as one flat
// sequence (a 1320-column row fills 330 of the 512 lane slots the per-row kernel gives it), four per thread with their
// loads in flight together, through accesses that do not keep the lines in the caches.
typedef float gather_f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void randomize_flat_kernel(float *y, const float *x, const int32_t *copy_from,
                                                             MatrixDim d_out, MatrixDim d_in, unsigned c4_per_row,
                                                             unsigned total4) {
  constexpr int U = 4;
  const unsigned nthreads = gridDim.x * 256;
  for (unsigned e = blockIdx.x * 256 + threadIdx.x; e < total4; e += nthreads * U) {
    gather_f4 v[U];
    size_t dst[U];
    bool on[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const unsigned eu = e + u * nthreads;
      on[u] = eu < total4;
      const unsigned r = on[u] ? eu / c4_per_row : 0, c = on[u] ? (eu - r * c4_per_row) * 4 : 0;
      const int src = copy_from[r];
      on[u] = on[u] && src >= 0 && src < d_in.rows;  // never read outside x
      dst[u] = (size_t)r * d_out.stride + c;
      if (on[u]) v[u] = __builtin_nontemporal_load(reinterpret_cast<const gather_f4 *>(x + (size_t)src * d_in.stride + c));
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (on[u]) __builtin_nontemporal_store(v[u], reinterpret_cast<gather_f4 *>(y + dst[u]));
  }
}

// y[r, k*D + c] = x[clamp(r + off[k]), c]
__global__ __launch_bounds__(256) void expand_kernel(float *y, const float *x, const int32_t *off, MatrixDim d_out,
                                                     MatrixDim d_in) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= d_out.cols) return;
  const int k = i / d_in.cols, src_col = i - k * d_in.cols;
  const int o = off[k];
  for (int r = blockIdx.y; r < d_out.rows; r += gridDim.y) {
    int src_row = r + o;
    if (src_row < 0) src_row = 0;
    if (src_row >= d_in.rows) src_row = d_in.rows - 1;
    y[(size_t)r * d_out.stride + i] = x[(size_t)src_row * d_in.stride + src_col];
  }
}

// y[r, c] = x[r, copy_from[c]], +inf for an index outside x
__global__ __launch_bounds__(256) void copy_kernel(float *y, const float *x, const int32_t *copy_from,
                                                   MatrixDim d_out, MatrixDim d_in) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= d_out.cols) return;
  const int src_col = copy_from[i];
  const bool ok = (src_col >= 0 && src_col < d_in.cols);
  for (int r = blockIdx.y; r < d_out.rows; r += gridDim.y)
    y[(size_t)r * d_out.stride + i] = ok ? x[(size_t)r * d_in.stride + src_col] : INFINITY;
}

dim3 grid2d(int cols, int rows) {
  int gx = (cols + 255) / 256;
  int gy = rows < 1 ? 1 : rows;
  int cap = 8192 / (gx > 0 ? gx : 1);
  if (gy > cap) gy = cap < 1 ? 1 : cap;
  return dim3(gx > 0 ? gx : 1, gy);
}

}  // namespace

KH_EXPORT int hipF_randomize(khStream_t s, float *y, const float *x, const int32_t *copy_from, MatrixDim d_out,
                             MatrixDim d_in) {
  KH_CHECK_INIT();
  KH_REQUIRE(kh::dim_ok(d_out) && kh::dim_ok(d_in) && d_out.cols == d_in.cols);
  if (d_out.rows == 0 || d_out.cols == 0) return KH_OK;
  KH_REQUIRE(y && x && copy_from);
  const bool vec = d_out.cols % 4 == 0 && d_out.stride % 4 == 0 && d_in.stride % 4 == 0 && kh::aligned16(y) &&
                   kh::aligned16(x);
  const unsigned long long total4 = (unsigned long long)d_out.rows * (d_out.cols / 4);
  if (vec && total4 >= (1ull << 20) && total4 < (1ull << 31)) {
    unsigned blocks = (unsigned)((total4 + 1023) / 1024);
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(randomize_flat_kernel, dim3(blocks), dim3(256), 0, kh::pick(s), y, x, copy_from, d_out, d_in,
                       (unsigned)(d_out.cols / 4), (unsigned)total4);
    return kh::launch_status();
  }
  const int grid = d_out.rows < 16384 ? d_out.rows : 16384;
  if (vec)
    hipLaunchKernelGGL((randomize_kernel<4>), dim3(grid), dim3(256), 0, kh::pick(s), y, x, copy_from, d_out, d_in);
  else
    hipLaunchKernelGGL((randomize_kernel<1>), dim3(grid), dim3(256), 0, kh::pick(s), y, x, copy_from, d_out, d_in);
  return kh::launch_status();
}

KH_EXPORT int hipF_expand(khStream_t s, float *y, const float *x, const int32_t *off, MatrixDim d_out,
                          MatrixDim d_in) {
  KH_CHECK_INIT();
  KH_REQUIRE(kh::dim_ok(d_out) && kh::dim_ok(d_in) && d_out.rows == d_in.rows);
  if (d_out.rows == 0 || d_out.cols == 0) return KH_OK;
  KH_REQUIRE(y && x && off && d_in.cols > 0 && d_out.cols % d_in.cols == 0);
  hipLaunchKernelGGL(expand_kernel, grid2d(d_out.cols, d_out.rows), dim3(256), 0, kh::pick(s), y, x, off, d_out,
                     d_in);
  return kh::launch_status();
}

KH_EXPORT int hipF_copy(khStream_t s, float *y, const float *x, const int32_t *copy_from, MatrixDim d_out,
                        MatrixDim d_in) {
  KH_CHECK_INIT();
  KH_REQUIRE(kh::dim_ok(d_out) && kh::dim_ok(d_in) && d_out.rows == d_in.rows);
  if (d_out.rows == 0 || d_out.cols == 0) return KH_OK;
  KH_REQUIRE(y && x && copy_from);
  hipLaunchKernelGGL(copy_kernel, grid2d(d_out.cols, d_out.rows), dim3(256), 0, kh::pick(s), y, x, copy_from, d_out,
                     d_in);
  return kh::launch_status();
}
