// Diagnostic: bytes per clock one CU pulls from L2-resident data with global_load_dwordx4, as a
// function of waves per CU and loads in flight per wave (every CU active, private 64 KiB regions or
// a region shared by the 4 workgroups of a "tile row").
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

template <int INFLIGHT>
__global__ void stream(const float4 *g, float *out, unsigned long long *cyc, int iters, int region_f4, int share) {
  const int blk = blockIdx.x / share;  // `share` consecutive workgroups read the same region
  const float4 *p = g + (size_t)blk * region_f4;
  float4 acc = make_float4(0, 0, 0, 0);
  const int nthr = blockDim.x;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  int off = threadIdx.x;
  for (int it = 0; it < iters; ++it) {
    float4 v[INFLIGHT];
#pragma unroll
    for (int k = 0; k < INFLIGHT; ++k) v[k] = p[(off + k * nthr) % region_f4];
#pragma unroll
    for (int k = 0; k < INFLIGHT; ++k) { acc.x += v[k].x; acc.y += v[k].y; acc.z += v[k].z; acc.w += v[k].w; }
    off = (off + INFLIGHT * nthr) % region_f4;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int INFLIGHT>
void run(int threads, int share, const float4 *g, float *out, unsigned long long *cyc) {
  const int nblk = 256, iters = 2000, region_f4 = 4096;  // 64 KiB per region
  stream<INFLIGHT><<<nblk, threads>>>(g, out, cyc, 200, region_f4, share);
  stream<INFLIGHT><<<nblk, threads>>>(g, out, cyc, iters, region_f4, share);
  (void)hipDeviceSynchronize();
  std::vector<unsigned long long> h(nblk);
  (void)hipMemcpy(h.data(), cyc, nblk * 8, hipMemcpyDeviceToHost);
  double c = 0;
  for (int i = 0; i < nblk; ++i) c += h[i];
  c /= nblk;
  const double bytes = (double)iters * INFLIGHT * threads * 16.0;
  printf("waves/CU %2d  loads in flight/wave %2d  shared by %d : %6.1f B/clk/CU\n", threads / 64, INFLIGHT, share, bytes / c);
}

int main() {
  float4 *g;
  float *out;
  unsigned long long *cyc;
  (void)hipMalloc(&g, (size_t)256 * 65536);
  (void)hipMemset(g, 0, (size_t)256 * 65536);
  (void)hipMalloc(&out, 256 * 1024 * 4);
  (void)hipMalloc(&cyc, 256 * 8);
  for (int share : {1, 4}) {
    run<4>(256, share, g, out, cyc);
    run<8>(256, share, g, out, cyc);
    run<16>(256, share, g, out, cyc);
    run<4>(512, share, g, out, cyc);
    run<8>(512, share, g, out, cyc);
    run<4>(1024, share, g, out, cyc);
    run<8>(1024, share, g, out, cyc);
  }
  return 0;
}
