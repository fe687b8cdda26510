// Diagnostic: cost of each ingredient of the 64x64x32 GEMM tile loop (16 dependent MFMAs per tile):
// LDS operand reads feeding the MFMAs, register->LDS stores, global loads, the per-tile barrier.
// One wave per SIMD on every CU, same instruction placement as the library kernel.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int PITCH = 36, TILE = 128 * PITCH;  // A(64 rows)+B(64 rows), 32 k + pad

template <bool READS, bool WRITES, bool LOADS, bool BARRIER, int NACC, int WKIND = 0, int DEPTH = 1, int RKIND = 0, int RPOS = 0, int LKIND = 0, int CLUSTER = 0>
__global__ __launch_bounds__(256) void probe(const float *__restrict__ g, float *out, unsigned long long *cyc, int tiles) {
  __shared__ __attribute__((aligned(16))) float lds[2 * TILE];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, li = lane & 31, lh = lane >> 5;
  for (int i = tid; i < 2 * TILE; i += 256) lds[i] = 0.001f * (i & 63);
  __syncthreads();
  f32x16 acc[NACC];
  for (int a = 0; a < NACC; ++a)
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  const int a_off = ((wid >> 1) * 32 + li) * PITCH + lh * 4;
  const int b_off = (64 + (wid & 1) * 32 + li) * PITCH + lh * 4;
  const int w_off0 = (tid >> 3) * PITCH + (tid & 7) * 4;
  const float *gp = g + (size_t)blockIdx.x * 4096 + tid * 4;
  float4 fa[2], fb[2], st[4], st2[4];
  fa[0] = *reinterpret_cast<const float4 *>(lds + a_off);
  fb[0] = *reinterpret_cast<const float4 *>(lds + b_off);
  fa[1] = fa[0];
  fb[1] = fb[0];
  for (int c = 0; c < 4; ++c) st[c] = *reinterpret_cast<const float4 *>(gp + c * 1024);
  for (int c = 0; c < 4; ++c) st2[c] = *reinterpret_cast<const float4 *>(gp + c * 1024 + 4096);
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(g + (size_t)blockIdx.x * 4096), 0, 0x7fffffff, 0x00020000);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int kt2 = 0; kt2 < tiles; kt2 += DEPTH)
#pragma unroll
  for (int kd = 0; kd < DEPTH; ++kd) {
    const int kt = kt2 + kd;
    float4 *stp = (DEPTH == 2 && kd == 1) ? st2 : st;
    float *cur = lds + (kt & 1) * TILE, *nxt = lds + ((kt + 1) & 1) * TILE;
    const float *gk = gp + (size_t)(kt & 3) * 1048576;  // a 16 MiB working set: L2 / Infinity Cache hits
#pragma unroll
    for (int gI = 0; gI < 4; ++gI) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float av = t == 0 ? fa[gI & 1].x : t == 1 ? fa[gI & 1].y : t == 2 ? fa[gI & 1].z : fa[gI & 1].w;
        const float bv = t == 0 ? fb[gI & 1].x : t == 1 ? fb[gI & 1].y : t == 2 ? fb[gI & 1].z : fb[gI & 1].w;
        acc[(gI * 4 + t) % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[(gI * 4 + t) % NACC], 0, 0, 0);
        if (BARRIER && gI == 3 && t == 0) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        // RPOS: 0 = A behind MFMA 0, B behind MFMA 1; 1 = both behind MFMA 0; 2 = behind MFMAs 2 and 3;
        //       3 = like 0 plus an early explicit wait behind MFMA 2
        if (RPOS == 3 && t == 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        for (int rr = 0; rr < 2; ++rr)
        if (READS && ((RPOS == 0 || RPOS == 3) ? (t == rr) : RPOS == 1 ? (t == 0) : (t == 2 + rr))) {
          const int t = rr;
          const float *src = (gI == 3 ? nxt : cur) + (t == 0 ? a_off : b_off) + ((gI + 1) & 3) * 8;
          float4 val;
          if (RKIND == 0) {
            val = *reinterpret_cast<const float4 *>(src);
          } else if (RKIND == 1) {
            const float2 lo = *reinterpret_cast<const float2 *>(src), hi = *reinterpret_cast<const float2 *>(src + 2);
            val = make_float4(lo.x, lo.y, hi.x, hi.y);
          } else {
            val = make_float4(src[0], src[1], src[2], src[3]);
          }
          if (t == 0) fa[(gI + 1) & 1] = val;
          else fb[(gI + 1) & 1] = val;
        }
        for (int c = 0; c < 4; ++c)
        if (CLUSTER == 0 ? ((gI == 1 || gI == 2) && (t == 0 || t == 2) && c == (gI - 1) * 2 + (t >> 1))
                         : (gI == 1 && t == 2)) {
          if (WKIND == 3) {
            // one 1 KiB piece per wave straight into LDS (unpadded image; addresses only matter for timing here)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gk + c * 1024),
                                             (__attribute__((address_space(3))) void *)(nxt + c * 1024 + wid * 256), 16, 0, 0);
          } else {
            float *dst = nxt + w_off0 + c * 32 * PITCH;
            if (WRITES && WKIND == 0) *reinterpret_cast<float4 *>(dst) = stp[c];
            if (WRITES && WKIND == 1) {
              *reinterpret_cast<float2 *>(dst) = make_float2(stp[c].x, stp[c].y);
              *reinterpret_cast<float2 *>(dst + 2) = make_float2(stp[c].z, stp[c].w);
            }
            if (WRITES && WKIND == 2) {
              dst[0] = stp[c].x; dst[1] = stp[c].y; dst[2] = stp[c].z; dst[3] = stp[c].w;
            }
            if (LOADS && LKIND == 0) stp[c] = *reinterpret_cast<const float4 *>(gk + c * 1024);
            if (LOADS && LKIND == 1) {
              typedef int v4i __attribute__((ext_vector_type(4)));
              const v4i r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(tid * 16 + c * 4096), (int)((kt & 3) * 4194304), 0);
              stp[c] = make_float4(__int_as_float(r.x), __int_as_float(r.y), __int_as_float(r.z), __int_as_float(r.w));
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = fa[0].x + fb[0].y + st[0].x + st[1].y + st[2].z + st[3].w + st2[0].x + st2[1].y + st2[2].z + st2[3].w;
  for (int a = 0; a < NACC; ++a)
    for (int r = 0; r < 16; ++r) s += acc[a][r];
  out[blockIdx.x * 256 + tid] = s;
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <bool READS, bool WRITES, bool LOADS, bool BARRIER, int NACC, int WKIND = 0, int DEPTH = 1, int RKIND = 0, int RPOS = 0, int LKIND = 0, int CLUSTER = 0>
void run(const char *name, const float *g, float *out, unsigned long long *cyc) {
  const int nblk = 256, tiles = 256;
  probe<READS, WRITES, LOADS, BARRIER, NACC, WKIND, DEPTH, RKIND, RPOS, LKIND, CLUSTER><<<nblk, 256>>>(g, out, cyc, tiles);
  probe<READS, WRITES, LOADS, BARRIER, NACC, WKIND, DEPTH, RKIND, RPOS, LKIND, CLUSTER><<<nblk, 256>>>(g, out, cyc, tiles);
  if (hipDeviceSynchronize() != hipSuccess) printf("launch failed\n");
  std::vector<unsigned long long> h(nblk);
  (void)hipMemcpy(h.data(), cyc, nblk * 8, hipMemcpyDeviceToHost);
  double sum = 0;
  for (int i = 0; i < nblk; ++i) sum += h[i];
  printf("%-46s %7.1f cycles/tile (16 MFMA = 1024)  %5.1f cycles/MFMA\n", name, sum / nblk / tiles, sum / nblk / tiles / 16);
}

int main() {
  float *g, *out;
  unsigned long long *cyc;
  (void)hipMalloc(&g, (size_t)64 * 1048576 * 4 + 256 * 4096 * 4 + 65536);
  (void)hipMemset(g, 0, (size_t)64 * 1048576 * 4 + 256 * 4096 * 4 + 65536);
  (void)hipMalloc(&out, 256 * 256 * 4);
  (void)hipMalloc(&cyc, 256 * 8);
  run<false, false, false, false, 1>("MFMA chain only", g, out, cyc);
  run<true, true, true, true, 1>("everything (b128 stores, global_load)", g, out, cyc);
  run<true, true, true, true, 1, 2>("everything, 4 x b32 stores", g, out, cyc);
  run<true, true, true, true, 1, 0, 1, 0, 0, 1>("everything, b128 stores, buffer_load", g, out, cyc);
  run<true, true, true, true, 1, 2, 1, 0, 0, 1>("everything, 4 x b32 stores, buffer_load", g, out, cyc);
  run<true, true, true, true, 1, 0, 1, 0, 0, 0, 1>("everything, b128, all copies in one gap", g, out, cyc);
  run<true, true, true, true, 1, 2, 1, 0, 0, 1, 1>("everything, b32, buffer_load, one gap", g, out, cyc);
  run<true, true, true, true, 1, 0, 1, 0, 1>("everything, b128, reads in one gap", g, out, cyc);
  return 0;
}
