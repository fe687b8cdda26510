// Diagnostic: the 64x64x32 tile loop with WHOLE-TILE operand fragments: all 8 LDS reads of the next
// tile in one MFMA gap (behind the barrier), copy units (LDS store + buffer load) in G gaps.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int v4i __attribute__((ext_vector_type(4)));
constexpr int PITCH = 36, TILE = 128 * PITCH;

// COPYGAPS: 4 = each copy unit in a gap of its own, 2 = two per gap, 1 = all four in one gap
// READGAP: MFMA index (0..15) behind which the barrier + 8 reads go
template <int COPYGAPS, int READGAP>
__global__ __launch_bounds__(256) void probe(const float *__restrict__ g, float *out, unsigned long long *cyc, int tiles) {
  __shared__ __attribute__((aligned(16))) float lds[2 * TILE];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, li = lane & 31, lh = lane >> 5;
  for (int i = tid; i < 2 * TILE; i += 256) lds[i] = 0.001f * (i & 63);
  __syncthreads();
  f32x16 acc = {0};
  const int a_off = ((wid >> 1) * 32 + li) * PITCH + lh * 4;
  const int b_off = (64 + (wid & 1) * 32 + li) * PITCH + lh * 4;
  const int w_off0 = (tid >> 3) * PITCH + (tid & 7) * 4;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(g + (size_t)blockIdx.x * 4096), 0, 0x7fffffff, 0x00020000);
  float4 fa[2][4], fb[2][4], st[4];
  for (int q = 0; q < 4; ++q) {
    fa[0][q] = *reinterpret_cast<const float4 *>(lds + a_off + q * 8);
    fb[0][q] = *reinterpret_cast<const float4 *>(lds + b_off + q * 8);
    fa[1][q] = fa[0][q];
    fb[1][q] = fb[0][q];
    const v4i r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, tid * 16 + q * 4096, 0, 0);
    st[q] = make_float4(__int_as_float(r.x), __int_as_float(r.y), __int_as_float(r.z), __int_as_float(r.w));
  }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int kt2 = 0; kt2 < tiles; kt2 += 2)
#pragma unroll
    for (int par = 0; par < 2; ++par) {
      const int kt = kt2 + par;
      float *nxt = lds + (par ^ 1) * TILE;
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        const float4 av = fa[par][m >> 2], bv = fb[par][m >> 2];
        const float a = (m & 3) == 0 ? av.x : (m & 3) == 1 ? av.y : (m & 3) == 2 ? av.z : av.w;
        const float b = (m & 3) == 0 ? bv.x : (m & 3) == 1 ? bv.y : (m & 3) == 2 ? bv.z : bv.w;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        // copy units: stores of tile kt+1 must precede the barrier
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int gap = COPYGAPS == 4 ? 1 + 2 * c : COPYGAPS == 2 ? 2 + 3 * (c >> 1) : 3;
          if (gap == m) {
            *reinterpret_cast<float4 *>(nxt + w_off0 + c * 32 * PITCH) = st[c];
            const v4i r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, tid * 16 + c * 4096, ((kt + 2) & 3) * 4194304, 0);
            st[c] = make_float4(__int_as_float(r.x), __int_as_float(r.y), __int_as_float(r.z), __int_as_float(r.w));
          }
        }
        if (m == READGAP) {
          asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            fa[par ^ 1][q] = *reinterpret_cast<const float4 *>(nxt + a_off + q * 8);
            fb[par ^ 1][q] = *reinterpret_cast<const float4 *>(nxt + b_off + q * 8);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = st[0].x + st[1].y + st[2].z + st[3].w;
  for (int r = 0; r < 16; ++r) s += acc[r];
  out[blockIdx.x * 256 + tid] = s;
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int COPYGAPS, int READGAP>
void run(const char *name, const float *g, float *out, unsigned long long *cyc) {
  const int nblk = 256, tiles = 256;
  probe<COPYGAPS, READGAP><<<nblk, 256>>>(g, out, cyc, tiles);
  probe<COPYGAPS, READGAP><<<nblk, 256>>>(g, out, cyc, tiles);
  if (hipDeviceSynchronize() != hipSuccess) printf("launch failed\n");
  std::vector<unsigned long long> h(nblk);
  (void)hipMemcpy(h.data(), cyc, nblk * 8, hipMemcpyDeviceToHost);
  double sum = 0;
  for (int i = 0; i < nblk; ++i) sum += h[i];
  printf("%-60s %7.1f cycles/tile (MFMA alone 1024)\n", name, sum / nblk / tiles);
}

int main() {
  float *g, *out;
  unsigned long long *cyc;
  (void)hipMalloc(&g, (size_t)64 * 1048576 * 4 + 256 * 4096 * 4 + 65536);
  (void)hipMemset(g, 0, (size_t)64 * 1048576 * 4 + 256 * 4096 * 4 + 65536);
  (void)hipMalloc(&out, 256 * 256 * 4);
  (void)hipMalloc(&cyc, 256 * 8);
  run<4, 10>("copies in 4 gaps (1,3,5,7), barrier+8 reads behind MFMA 10", g, out, cyc);
  run<4, 12>("copies in 4 gaps, barrier+8 reads behind MFMA 12", g, out, cyc);
  run<4, 14>("copies in 4 gaps, barrier+8 reads behind MFMA 14", g, out, cyc);
  run<2, 10>("copies in 2 gaps (2,5), barrier+8 reads behind MFMA 10", g, out, cyc);
  run<1, 10>("copies in 1 gap (3), barrier+8 reads behind MFMA 10", g, out, cyc);
  return 0;
}
