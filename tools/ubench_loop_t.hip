// The steady-state sweep of gemv_k256t_kernel in isolation (gfx950 / MI355X), next to the loop of
// gemv_k256m_kernel: index words from a per-lane counter instead of HBM, no hand-over, no stores.
// Per index-wave (one index for 64 lanes): new = 2 v_perm + 2 v_xor + 4 ds_read_b64_tr_b16 + 1/2
// ds_read_b128 + 2 v_mfma_16x16x32; old = 4 v_perm + 2 ds_read_b128 + 4 v_mfma_4x4x4.
//   hipcc --offload-arch=gfx950 -O3 -I vptq_amd/csrc tools/ubench_loop_t.hip -o tools/_build/ubench_loop_t
// Reported: SIMD cycles (s_memtime) per index-wave with 4 (or 2) waves sharing the SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "common.h"
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
using namespace vptq;
typedef _Float16 h8v_t __attribute__((ext_vector_type(8)));
enum { RD = 1, MF = 2, XR = 4, PLAIN = 8, AH2 = 16, OLD = 32 };

static __device__ __forceinline__ u32x2 lds_tr8(uint32_t a) {
  typedef __attribute__((address_space(3))) s4_t lds_s4_t;
  return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_t*)(uintptr_t)a));
}
static __device__ __forceinline__ u32x2 lds_ld8(uint32_t a) {
  typedef __attribute__((address_space(3))) u32x2 lds_u2_t;
  return *(const lds_u2_t*)(uintptr_t)a;
}

template <int MODE, int THREADS, int AH = 2>
__global__ __launch_bounds__(THREADS) void loop_kernel(float* out, unsigned long long* cyc, int iters, unsigned seed) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 16384 + 1024; i += THREADS) ((unsigned*)smem)[i] = 0x2c002c00u + i;
  __syncthreads();
  uint32_t r = (tid * 2654435761u + seed) ^ (blockIdx.x * 40503u);
  unsigned long long t0, t1;
  float res = 0.f;
  if constexpr (MODE & OLD) {
    const uint32_t hi = (lane >> 3) & 1u;
    const uint32_t baseA = ((hi << 3) | (lane & 7u)) << 4;
    const uint32_t baseB = (((hi ^ 1u) << 3) | (lane & 7u)) << 4;
    const uint32_t selGA[2] = {0x0c0c0400u | (hi << 8), 0x0c0c0600u | (hi << 8)};
    const uint32_t selGB[2] = {0x0c0c0400u | ((hi ^ 1u) << 8), 0x0c0c0600u | ((hi ^ 1u) << 8)};
    const int j = lane & 3;
    const uint32_t selA[2] = {j == 0 ? 0x0c0c0504u : j == 1 ? 0x05040c0cu : 0x0c0c0c0cu, j == 0 ? 0x0c0c0706u : j == 1 ? 0x07060c0cu : 0x0c0c0c0cu};
    const uint32_t selB[2] = {j == 2 ? 0x0c0c0504u : j == 3 ? 0x05040c0cu : 0x0c0c0c0cu, j == 2 ? 0x0c0c0706u : j == 3 ? 0x07060c0cu : 0x0c0c0c0cu};
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int it = 0; it < iters; ++it) {
      u32x4 words;
#pragma unroll
      for (int q = 0; q < 4; ++q) { r = r * 1664525u + 1013904223u; words[q] = r; }
      const u32x4 xq = lds_load16(65536u + (uint32_t)(wave * 16 + (lane >> 2)) * 16u);
      u32x4 cv[AH + 1], rv[AH + 1];
      auto gather = [&](int u) {
        const uint32_t w = words[u >> 1];
        cv[u % (AH + 1)] = lds_load16(__builtin_amdgcn_perm(w, baseA, selGA[u & 1]));
        rv[u % (AH + 1)] = lds_load16(__builtin_amdgcn_perm(w, baseB, selGB[u & 1]));
      };
#pragma unroll
      for (int u = 0; u < AH; ++u) gather(u);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        __builtin_amdgcn_sched_barrier(0);
        if (u + AH < 8) gather(u + AH);
        __builtin_amdgcn_sched_barrier(0);
        const u32x4 c = cv[u % (AH + 1)], rr = rv[u % (AH + 1)];
        const u32x2 xo = u32x2{__builtin_amdgcn_perm(xq[u >> 1], 0u, selA[u & 1]), __builtin_amdgcn_perm(xq[u >> 1], 0u, selB[u & 1])};
        acc0 = F16::mfma4(xo, u32x2{c[0], c[1]}, acc0);
        acc1 = F16::mfma4(xo, u32x2{c[2], c[3]}, acc1);
        acc0 = F16::mfma4(xo, u32x2{rr[0], rr[1]}, acc0);
        acc1 = F16::mfma4(xo, u32x2{rr[2], rr[3]}, acc1);
      }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) : "v"(acc0[0] + acc1[0]) : "memory");
    res = acc0[0] + acc1[1];
  } else {
    const uint32_t kg = (uint32_t)lane >> 4, s16 = (uint32_t)lane & 15u;
    const uint32_t src_e = s16 >> 2, src_c = s16 & 3u;
    const uint32_t rot_b = kg & 1u, rot_h = (MODE & XR) ? (src_c & 1u) : 0u;
    const uint32_t rep = (src_e << 1) | (src_c >> 1);
    const uint32_t baseX = (rot_b << 7) | (rep << 4) | (rot_h << 3);
    const uint32_t baseY = ((rot_b ^ 1u) << 7) | (rep << 4) | (rot_h << 3);
    const uint32_t selX[2] = {0x0c020000u | ((4u + rot_b) << 8), 0x0c020000u | ((6u + rot_b) << 8)};
    const uint32_t selY[2] = {0x0c020000u | ((4u + (rot_b ^ 1u)) << 8), 0x0c020000u | ((6u + (rot_b ^ 1u)) << 8)};
    const uint32_t xa_addr = 65536u + (uint32_t)wave * 256u + (kg << 4);
    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int it = 0; it < iters; ++it) {
      u32x4 words;
#pragma unroll
      for (int q = 0; q < 4; ++q) { r = r * 1664525u + 1013904223u; words[q] = r; }
      constexpr int NB = (MODE & AH2) ? 3 : 2;
      u32x2 g[NB][2][2][2];
      u32x4 xa[NB];
      auto gather_pair = [&](int p) {
        uint32_t w = words[p];
        asm volatile("" : "+v"(w));
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const uint32_t aX = __builtin_amdgcn_perm(w, baseX, selX[t]);
          const uint32_t aY = __builtin_amdgcn_perm(w, baseY, selY[t]);
          const uint32_t aX2 = (MODE & XR) ? (aX ^ 8u) : (aX + 8u), aY2 = (MODE & XR) ? (aY ^ 8u) : (aY + 8u);
          if constexpr (!(MODE & RD)) {
            asm volatile("" :: "v"(aX2), "v"(aY2));
            g[p % NB][t][0][0] = u32x2{w, aX}; g[p % NB][t][0][1] = u32x2{aX, w};
            g[p % NB][t][1][0] = u32x2{w, aY}; g[p % NB][t][1][1] = u32x2{aY, w};
          } else if constexpr (MODE & PLAIN) {
            g[p % NB][t][0][0] = lds_ld8(aX); g[p % NB][t][0][1] = lds_ld8(aX2);
            g[p % NB][t][1][0] = lds_ld8(aY); g[p % NB][t][1][1] = lds_ld8(aY2);
          } else {
            g[p % NB][t][0][0] = lds_tr8(aX); g[p % NB][t][0][1] = lds_tr8(aX2);
            g[p % NB][t][1][0] = lds_tr8(aY); g[p % NB][t][1][1] = lds_tr8(aY2);
          }
        }
        xa[p % NB] = lds_load16(xa_addr + (uint32_t)p * 64u);
      };
      constexpr int AHEAD = (MODE & AH2) ? 2 : 1;
#pragma unroll
      for (int p = 0; p < AHEAD; ++p) gather_pair(p);
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        __builtin_amdgcn_sched_barrier(0);
        if (p + AHEAD < 4) gather_pair(p + AHEAD);
        __builtin_amdgcn_sched_barrier(0);
        const u32x4 a = xa[p % NB];
#pragma unroll
        for (int rr = 0; rr < 2; ++rr)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const u32x2 t0v = g[p % NB][0][rr][h], t1v = g[p % NB][1][rr][h];
            const u32x4 b = u32x4{t0v[0], t0v[1], t1v[0], t1v[1]};
            if constexpr (MODE & MF)
              acc[h] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8v_t, a), __builtin_bit_cast(h8v_t, b), acc[h], 0, 0, 0);
            else asm volatile("" :: "v"(a), "v"(b));
          }
      }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) : "v"(acc[0][0] + acc[1][0]) : "memory");
    res = acc[0][0] + acc[1][1];
  }
  if (res == 1234.5f) out[tid] = res;
  if (lane == 0) cyc[blockIdx.x * (THREADS / 64) + wave] = t1 - t0;
}

template <int MODE, int THREADS, int AH = 2>
static void run(const char* name, float* out, unsigned long long* cyc) {
  const int iters = 400, wgs = 256;
  auto kern = loop_kernel<MODE, THREADS, AH>;
  CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kern, dim3(wgs), dim3(THREADS), 72 * 1024, 0, out, cyc, iters, 1u);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(kern, dim3(wgs), dim3(THREADS), 72 * 1024, 0, out, cyc, iters, 7u);
  CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  const int nw = wgs * THREADS / 64;
  unsigned long long* h = (unsigned long long*)malloc(nw * 8);
  CHECK(hipMemcpy(h, cyc, nw * 8, hipMemcpyDeviceToHost));
  double sum = 0; for (int i = 0; i < nw; ++i) sum += (double)h[i];
  const double per_wave = sum / nw;                       // shader clocks for iters steps of 8 index-waves
  const double waves_per_simd = THREADS / 64 / 4.0;
  printf("%-64s %7.1f cycles per index-wave and SIMD (%4.0f per step and wave; %.0f us wall)\n", name,
         per_wave / (iters * 8.0 * waves_per_simd), per_wave / iters, ms * 1e3);
  free(h);
}

int main() {
  float* out; unsigned long long* cyc;
  CHECK(hipMalloc(&out, 4096 * 4)); CHECK(hipMalloc(&cyc, 8192 * 8));
  run<OLD, 1024>("old loop (4x4x4 MFMA, b128 gathers), 4 waves / SIMD", out, cyc);
  run<RD | MF | XR, 1024>("new loop: tr reads (rotated chunks) + 16x16x32 MFMA", out, cyc);
  run<RD | MF, 1024>("new loop, chunks not rotated (+8 instead of ^8)", out, cyc);
  run<RD | XR, 1024>("new loop, no MFMA", out, cyc);
  run<MF | XR, 1024>("new loop, no LDS gathers", out, cyc);
  run<XR, 1024>("new loop, neither", out, cyc);
  run<RD | MF | XR | PLAIN, 1024>("new loop, plain ds_read_b64 instead of the transposing read", out, cyc);
  run<RD | XR | PLAIN, 1024>("  ... no MFMA", out, cyc);
  run<RD | MF | XR | AH2, 1024>("new loop, gathers two tile pairs ahead", out, cyc);
  run<RD | MF | XR, 512>("new loop, 2 waves / SIMD", out, cyc);
  run<OLD, 512>("old loop, 2 waves / SIMD", out, cyc);
  // how the 4x4x4 loop depends on the gathers in flight per SIMD (waves x lookahead): latency or throughput?
  run<OLD, 1024, 1>("old loop, 4 waves / SIMD, lookahead 1", out, cyc);
  run<OLD, 1024, 3>("old loop, 4 waves / SIMD, lookahead 3", out, cyc);
  run<OLD, 1024, 4>("old loop, 4 waves / SIMD, lookahead 4", out, cyc);
  run<OLD, 512, 3>("old loop, 2 waves / SIMD, lookahead 3", out, cyc);
  run<OLD, 512, 4>("old loop, 2 waves / SIMD, lookahead 4", out, cyc);
  run<OLD, 512, 6>("old loop, 2 waves / SIMD, lookahead 6", out, cyc);
  run<OLD, 256, 2>("old loop, 1 wave / SIMD, lookahead 2", out, cyc);
  run<OLD, 256, 4>("old loop, 1 wave / SIMD, lookahead 4", out, cyc);
  run<OLD, 256, 6>("old loop, 1 wave / SIMD, lookahead 6", out, cyc);
  run<OLD, 256, 7>("old loop, 1 wave / SIMD, lookahead 7", out, cyc);
  return 0;
}
