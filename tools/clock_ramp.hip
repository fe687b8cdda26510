// Diagnostic: the shader clock an MFMA-dense kernel gets, launch by launch, from an idle GPU.
// Every launch (fp32 MFMAs on every SIMD, ~0.25 ms) stamps s_memtime / s_memrealtime (100 MHz) around its
// loop; clock = cycles / real time.  usage: clock_ramp [launches=240] [idle_ms=500]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void burn(float *out, unsigned long long *stamps, int iters, int slot) {
  f32x16 acc0 = {0}, acc1 = {0};
  const float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f + 1.0f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc1, 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    stamps[2 * slot] = t1 - t0;
    stamps[2 * slot + 1] = r1 - r0;
  }
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
  if (s == 12345.678f) out[0] = s;  // keeps the MFMAs alive
}

int main(int argc, char **argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 240;
  const int idle_ms = argc > 2 ? atoi(argv[2]) : 500;
  float *out;
  unsigned long long *stamps;
  if (hipMalloc(&out, 4) != hipSuccess || hipMalloc(&stamps, n * 16) != hipSuccess) return 1;
  hipStream_t st;
  (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  for (int pass = 0; pass < 2; ++pass) {
    (void)hipMemsetAsync(stamps, 0, n * 16, st);
    (void)hipStreamSynchronize(st);
    usleep(idle_ms * 1000);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, st);
    for (int i = 0; i < n; ++i) burn<<<256, 256, 0, st>>>(out, stamps, 4000, i);  // 8000 MFMAs x 64 cycles ~ 0.21 ms at 2.4 GHz
    (void)hipEventRecord(e1, st);
    if (hipStreamSynchronize(st) != hipSuccess) return 2;
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(2 * n);
    (void)hipMemcpy(h.data(), stamps, n * 16, hipMemcpyDeviceToHost);
    printf("pass %d (%d ms idle before): %d launches in %.2f ms; GHz per launch, then ms since start:\n", pass, idle_ms, n, ms);
    double t = 0.0;
    for (int i = 0; i < n; ++i) {
      const double ghz = (double)h[2 * i] / (double)h[2 * i + 1] * 0.1;
      t += (double)h[2 * i + 1] * 1e-5;
      if (i < 40 || i % 10 == 0) printf("  #%d %.3f GHz @%.2f ms\n", i, ghz, t);
    }
  }
  return 0;
}
