// Diagnostic: time per kernel of a chain of N small dependent kernels launched (a) one by one on a
// stream, (b) as one captured hipGraph; and the same for a chain that alternates a ~20 us kernel
// with a tiny one (the shape of the training step).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>

__global__ void tiny(float *p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.0f; }
__global__ void busy(float *p, int iters) {
  float v = p[blockIdx.x * 256 + threadIdx.x];
  for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
  p[blockIdx.x * 256 + threadIdx.x] = v;
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
  float *d;
  (void)hipMalloc(&d, 256 * 256 * 4);
  (void)hipMemset(d, 0, 256 * 256 * 4);
  hipStream_t s;
  (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  const int N = 50, reps = 200;
  for (int mode = 0; mode < 2; ++mode) {
    auto chain = [&](hipStream_t st) {
      for (int i = 0; i < N; ++i) {
        if (mode == 1 && (i & 1) == 0) busy<<<256, 256, 0, st>>>(d, 6000);
        else tiny<<<1, 64, 0, st>>>(d);
      }
    };
    for (int w = 0; w < 5; ++w) chain(s);
    (void)hipStreamSynchronize(s);
    double t0 = now();
    for (int r = 0; r < reps; ++r) chain(s);
    (void)hipStreamSynchronize(s);
    const double t_stream = (now() - t0) / reps / N * 1e6;
    hipGraph_t g;
    hipGraphExec_t ge;
    (void)hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    chain(s);
    (void)hipStreamEndCapture(s, &g);
    (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int w = 0; w < 5; ++w) (void)hipGraphLaunch(ge, s);
    (void)hipStreamSynchronize(s);
    t0 = now();
    for (int r = 0; r < reps; ++r) (void)hipGraphLaunch(ge, s);
    (void)hipStreamSynchronize(s);
    const double t_graph = (now() - t0) / reps / N * 1e6;
    printf("%s: stream %.2f us per kernel, graph %.2f us per kernel\n", mode ? "busy/tiny alternating" : "tiny kernels only", t_stream, t_graph);
  }
  return 0;
}
