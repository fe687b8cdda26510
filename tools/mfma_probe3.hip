// Diagnostic: sustained fp32 MFMA rate and the clock the chip holds for the two fp32 shapes
// (32x32x2 and 16x16x4), on random data, every SIMD busy for ~0.3 s.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(256) void probe(const float *in, float *out, unsigned long long *cyc, int iters) {
  const float a0 = in[threadIdx.x], b0 = in[256 + threadIdx.x], a1 = in[512 + threadIdx.x], b1 = in[768 + threadIdx.x];
  f32x16 acc = {0};
  f32x4 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (SHAPE == 32) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32((u & 1) ? a1 : a0, (u & 2) ? b1 : b0, acc, 0, 0, 0);
      } else {
        // two 16x16x4 per 32x32x2-equivalent of work (same FLOPs: 2 x 2048)
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32((u & 1) ? a1 : a0, (u & 2) ? b1 : b0, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32((u & 1) ? a0 : a1, (u & 2) ? b1 : b0, c1, 0, 0, 0);
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
  for (int r = 0; r < 16; ++r) s += acc[r];
  for (int r = 0; r < 4; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) {
    cyc[blockIdx.x * 2] = t1 - t0;
    cyc[blockIdx.x * 2 + 1] = r1 - r0;
  }
}

template <int SHAPE>
void run(const char *name, const float *in, float *out, unsigned long long *cyc) {
  const int nblk = 256, iters = 400000;  // ~0.2 s
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  probe<SHAPE><<<nblk, 256>>>(in, out, cyc, 1000);
  (void)hipEventRecord(e0);
  probe<SHAPE><<<nblk, 256>>>(in, out, cyc, iters);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(nblk * 2);
  (void)hipMemcpy(h.data(), cyc, nblk * 16, hipMemcpyDeviceToHost);
  double c = 0, r = 0;
  for (int i = 0; i < nblk; ++i) { c += h[2 * i]; r += h[2 * i + 1]; }
  const double flops = 2.0 * 32 * 32 * 2 * 16.0 * iters * 4 * nblk;
  printf("%-12s %7.1f TFLOP/s  %6.2f cycles per 4096 FLOP/SIMD  in-kernel clock %.2f GHz  (%.0f ms)\n", name,
         flops / (ms * 1e-3) / 1e12, c / nblk / (16.0 * iters), c / r * 0.1, ms);
}

int main() {
  float *in, *out;
  unsigned long long *cyc;
  (void)hipMalloc(&in, 4096);
  std::vector<float> h(1024);
  srand(1);
  for (auto &v : h) v = (rand() / (float)RAND_MAX - 0.5f) * 1e-3f;
  (void)hipMemcpy(in, h.data(), 4096, hipMemcpyHostToDevice);
  (void)hipMalloc(&out, 256 * 256 * 4);
  (void)hipMalloc(&cyc, 256 * 16);
  for (int rep = 0; rep < 2; ++rep) {
    run<32>("32x32x2", in, out, cyc);
    run<16>("2x 16x16x4", in, out, cyc);
  }
  return 0;
}
