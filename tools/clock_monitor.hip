// Diagnostic: samples the shader clock (s_memtime ticks per s_memrealtime tick x 100 MHz) every 10 ms
// from one sleeping wave, for `seconds`; run it beside another process to see the clock that
// process's kernels actually get.  usage: clock_monitor [seconds]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

__global__ void monitor(float *ghz, int samples) {
  if (threadIdx.x != 0) return;
  for (int s = 0; s < samples; ++s) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long r1 = r0;
    while (r1 - r0 < 1000000ull) {  // 10 ms at 100 MHz
      __builtin_amdgcn_s_sleep(127);
      r1 = __builtin_amdgcn_s_memrealtime();
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    ghz[s] = (float)((double)(t1 - t0) / (double)(r1 - r0) * 0.1);
  }
}

int main(int argc, char **argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 3.0;
  const int samples = (int)(seconds * 100);
  float *d;
  (void)hipMalloc(&d, samples * 4);
  monitor<<<1, 64>>>(d, samples);
  if (hipDeviceSynchronize() != hipSuccess) return 1;
  std::vector<float> h(samples);
  (void)hipMemcpy(h.data(), d, samples * 4, hipMemcpyDeviceToHost);
  for (int i = 0; i < samples; i += 10) {
    printf("t=%4.1fs ", i * 0.01);
    for (int j = i; j < i + 10 && j < samples; ++j) printf(" %.2f", h[j]);
    printf("\n");
  }
  return 0;
}
