// Diagnostic: cycles per v_mfma_f32_32x32x2_f32 for a dependent chain and for 2/4 independent
// accumulators, alone and with LDS reads in the gaps; one wave per SIMD on every CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, int READS>
__global__ __launch_bounds__(256) void probe(float *out, unsigned long long *cyc, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = 0.001f * (i & 63);
  __syncthreads();
  f32x16 acc[NACC];
  for (int a = 0; a < NACC; ++a)
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  float4 f0 = *reinterpret_cast<float4 *>(lds + threadIdx.x * 4);
  float4 f1 = *reinterpret_cast<float4 *>(lds + threadIdx.x * 4 + 1024);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  float4 t[16];
  for (int u = 0; u < 16; ++u) t[u] = make_float4(0.f, 0.f, 0.f, 0.f);
  const float *rp = lds + ((threadIdx.x * 4) & 1020);
  for (int it = 0; it < iters; ++it) {
    // values read during the previous pass are consumed here, long after their reads were issued
    float4 s4 = t[0];
#pragma unroll
    for (int u = 1; u < 16; ++u) s4.x += t[u].x;
    f0.y += s4.x;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      acc[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(f0.x, f1.x, acc[u % NACC], 0, 0, 0);
      if (READS > 0 && (u % (16 / READS)) == 0) t[u] = *reinterpret_cast<const float4 *>(rp + u * 256 + (it & 1) * 4096);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = f0.y;
  for (int a = 0; a < NACC; ++a)
    for (int r = 0; r < 16; ++r) s += acc[a][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NACC, int READS>
void run(const char *name, float *out, unsigned long long *cyc, int nblk) {
  const int iters = 200;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  probe<NACC, READS><<<nblk, 256>>>(out, cyc, iters);
  hipEventRecord(e0);
  probe<NACC, READS><<<nblk, 256>>>(out, cyc, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(nblk);
  hipMemcpy(h.data(), cyc, nblk * 8, hipMemcpyDeviceToHost);
  double sum = 0;
  for (int i = 0; i < nblk; ++i) sum += h[i];
  const double per = sum / nblk / (iters * 16.0);
  const double tf = 2.0 * 32 * 32 * 2 * 16.0 * iters * 4 * nblk / (ms * 1e-3) / 1e12;
  printf("%-34s %6.1f cycles/MFMA   %6.1f TFLOP/s (launch incl.)  clock~%.2f GHz\n", name, per, tf,
         sum / nblk / (ms * 1e-3) / 1e9);
}

int main() {
  float *out;
  unsigned long long *cyc;
  const int nblk = 256;
  hipMalloc(&out, nblk * 256 * 4);
  hipMalloc(&cyc, nblk * 8);
  run<1, 0>("1 acc (dependent chain)", out, cyc, nblk);
  run<2, 0>("2 acc", out, cyc, nblk);
  run<4, 0>("4 acc", out, cyc, nblk);
  run<1, 8>("1 acc + 8 ds_read_b128 / 16 MFMA", out, cyc, nblk);
  run<4, 8>("4 acc + 8 ds_read_b128 / 16 MFMA", out, cyc, nblk);
  run<4, 16>("4 acc + 16 ds_read_b128 / 16 MFMA", out, cyc, nblk);
  return 0;
}
