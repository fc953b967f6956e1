// The steady-state inner loop of gemv_k256m_kernel in isolation (gfx950 / MI355X): per index
// 2 v_perm_b32 (gather addresses) + 2 ds_read_b128 (conflict-free lane-split image) + 4
// v_mfma_f32_4x4x4_16b_f16, gathers two indices ahead, fenced like the kernel.  Which parts
// overlap?  Each variant leaves one ingredient out; index words come from a per-lane counter
// instead of HBM.  1024-thread workgroups, one per CU (4 waves per SIMD), 64 KiB LDS image.
//   hipcc --offload-arch=gfx950 -O3 -I vptq_amd/csrc tools/ubench_loop.hip -o /tmp/ubench_loop && /tmp/ubench_loop
// Reported: SIMD cycles per index-wave (one index for 64 lanes), 4 waves sharing the SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#include "common.h"

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

using namespace vptq;

enum { LDS = 1, MFMA = 2, PERM = 4 };

template <int MODE, int AHEAD>
__global__ __launch_bounds__(1024) void loop_kernel(float* out, int iters, unsigned seed) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 16384; i += 1024) ((unsigned*)smem)[i] = 0x2c002c00u + i;  // small f16 pairs
  __syncthreads();
  const uint32_t hi = (lane >> 3) & 1u;
  const uint32_t baseA = ((hi << 3) | (lane & 7u)) << 4;
  const uint32_t baseB = (((hi ^ 1u) << 3) | (lane & 7u)) << 4;
  const uint32_t selGA[2] = {0x0c0c0400u | (hi << 8), 0x0c0c0600u | (hi << 8)};
  const uint32_t selGB[2] = {0x0c0c0400u | ((hi ^ 1u) << 8), 0x0c0c0600u | ((hi ^ 1u) << 8)};
  uint32_t r = (tid * 2654435761u + seed) ^ (blockIdx.x * 40503u);
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  u32x2 xo = u32x2{0x3c000000u + (uint32_t)lane, 0u};
  u32x4 fixed_c = u32x4{r, r + 1, r + 2, r + 3}, fixed_r = u32x4{r + 4, r + 5, r + 6, r + 7};
  for (int it = 0; it < iters; ++it) {
    u32x4 words;
#pragma unroll
    for (int q = 0; q < 4; ++q) { r = r * 1664525u + 1013904223u; words[q] = r; }
    u32x4 cv[AHEAD + 1], rv[AHEAD + 1];
    auto gather = [&](int u) {
      const uint32_t w = words[u >> 1];
      const int h = u & 1;
      uint32_t aC, aR;
      if (MODE & PERM) {
        aC = __builtin_amdgcn_perm(w, baseA, selGA[h]);
        aR = __builtin_amdgcn_perm(w, baseB, selGB[h]);
      } else {
        aC = baseA + ((uint32_t)(u * 37 + it) & 255u) * 256u;   // wave-uniform row: still conflict free
        aR = baseB + ((uint32_t)(u * 91 + it) & 255u) * 256u;
      }
      if (MODE & LDS) {
        cv[u % (AHEAD + 1)] = lds_load16(aC);
        rv[u % (AHEAD + 1)] = lds_load16(aR);
      } else {
        cv[u % (AHEAD + 1)] = fixed_c; rv[u % (AHEAD + 1)] = fixed_r;
        cv[u % (AHEAD + 1)][0] ^= aC; rv[u % (AHEAD + 1)][0] ^= aR;   // keeps the address arithmetic alive
      }
    };
#pragma unroll
    for (int u = 0; u < AHEAD; ++u) gather(u);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      __builtin_amdgcn_sched_barrier(0);
      if (u + AHEAD < 8) gather(u + AHEAD);
      __builtin_amdgcn_sched_barrier(0);
      const u32x4 c = cv[u % (AHEAD + 1)], rr = rv[u % (AHEAD + 1)];
      if (MODE & MFMA) {
        acc0 = F16::mfma4(xo, u32x2{c[0], c[1]}, acc0);
        acc1 = F16::mfma4(xo, u32x2{c[2], c[3]}, acc1);
        acc0 = F16::mfma4(xo, u32x2{rr[0], rr[1]}, acc0);
        acc1 = F16::mfma4(xo, u32x2{rr[2], rr[3]}, acc1);
      } else {
        asm volatile("" :: "v"(c), "v"(rr));   // the data must have arrived, nothing is computed
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  if (acc0[0] + acc1[0] + acc0[3] == 1234.5f) out[blockIdx.x * 1024 + tid] = acc0[1];
}

// The 2-4-token form of the loop (sweep_tokens in gemv_k256m.hip): per PAIR of indices 4 address
// perms + 4 ds_read_b128, 16 transposing perms and 8 MFMAs into 8 accumulators.
template <int MODE>
__global__ __launch_bounds__(1024) void loop_tokens_kernel(float* out, int iters, unsigned seed) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 16384; i += 1024) ((unsigned*)smem)[i] = 0x2c002c00u + i;
  __syncthreads();
  const uint32_t hi = (lane >> 3) & 1u;
  const uint32_t baseA = ((hi << 3) | (lane & 7u)) << 4;
  const uint32_t baseB = (((hi ^ 1u) << 3) | (lane & 7u)) << 4;
  const uint32_t selGA[2] = {0x0c0c0400u | (hi << 8), 0x0c0c0600u | (hi << 8)};
  const uint32_t selGB[2] = {0x0c0c0400u | ((hi ^ 1u) << 8), 0x0c0c0600u | ((hi ^ 1u) << 8)};
  uint32_t r = (tid * 2654435761u + seed) ^ (blockIdx.x * 40503u);
  f32x4 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const u32x4 xq = u32x4{0x3c003c00u + (uint32_t)lane, 0x38003800u, 0x34003400u, 0x30003000u};
  for (int it = 0; it < iters; ++it) {
    u32x4 words;
#pragma unroll
    for (int q = 0; q < 4; ++q) { r = r * 1664525u + 1013904223u; words[q] = r; }
    u32x4 ga[2][2], gb[2][2];
    auto gather_pair = [&](int p) {
      const uint32_t w = words[p];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        ga[p & 1][h] = lds_load16(__builtin_amdgcn_perm(w, baseA, selGA[h]));
        gb[p & 1][h] = lds_load16(__builtin_amdgcn_perm(w, baseB, selGB[h]));
      }
    };
    gather_pair(0);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      __builtin_amdgcn_sched_barrier(0);
      if (p + 1 < 4) gather_pair(p + 1);
      __builtin_amdgcn_sched_barrier(0);
      const u32x2 xa = u32x2{xq[p], xq[p]};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint32_t a0 = ga[p & 1][0][q], a1 = ga[p & 1][1][q];
        const uint32_t b0 = gb[p & 1][0][q], b1 = gb[p & 1][1][q];
        const u32x2 wlo = u32x2{__builtin_amdgcn_perm(a1, a0, 0x05040100u), __builtin_amdgcn_perm(b1, b0, 0x05040100u)};
        const u32x2 whi = u32x2{__builtin_amdgcn_perm(a1, a0, 0x07060302u), __builtin_amdgcn_perm(b1, b0, 0x07060302u)};
        if (MODE & MFMA) {
          acc[2 * q] = F16::mfma4(xa, wlo, acc[2 * q]);
          acc[2 * q + 1] = F16::mfma4(xa, whi, acc[2 * q + 1]);
        } else {
          asm volatile("" :: "v"(wlo), "v"(whi));
        }
        if (MODE & PERM) __builtin_amdgcn_sched_barrier(0);   // PERM bit: fenced per step like the kernel
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
  if (s == 1234.5f) out[blockIdx.x * 1024 + tid] = s;
}

template <int MODE>
static void run_tokens(const char* name, int iters, double ghz) {
  const int blocks = 256;
  float* out; CHECK(hipMalloc(&out, (size_t)blocks * 1024 * 4));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  auto kern = loop_tokens_kernel<MODE>;
  CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(1024), 65536, 0, out, iters / 8, 1u);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(1024), 65536, 0, out, iters, 3u);
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double ns = ms * 1e6 / (4.0 * iters * 8);
  printf("%-52s %8.3f ms  %6.2f ns = %6.1f SIMD cycles per index-wave @%.1f GHz\n", name, ms, ns, ns * ghz, ghz);
  CHECK(hipFree(out));
}

template <int MODE, int AHEAD>
static void run(const char* name, int iters, double ghz) {
  const int blocks = 256;
  float* out; CHECK(hipMalloc(&out, (size_t)blocks * 1024 * 4));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  auto kern = loop_kernel<MODE, AHEAD>;
  CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(1024), 65536, 0, out, iters / 8, 1u);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(1024), 65536, 0, out, iters, 3u);
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double index_waves_per_simd = 4.0 * iters * 8;   // 4 waves per SIMD, 8 indices per sweep
  const double ns = ms * 1e6 / index_waves_per_simd;
  printf("%-44s ahead=%d %8.3f ms  %6.2f ns = %6.1f SIMD cycles per index-wave @%.1f GHz\n", name, AHEAD, ms, ns,
         ns * ghz, ghz);
  CHECK(hipFree(out));
}

int main() {
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
  printf("%s CUs=%d clock=%d kHz; 1024-thread workgroups, 1 per CU\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
  const int it = 20000;
  run<LDS | MFMA | PERM, 2>("full: 2 perm + 2 ds_read_b128 + 4 MFMA", it, 2.4);
  run<LDS | MFMA | PERM, 3>("full", it, 2.4);
  run<LDS | MFMA | PERM, 4>("full", it, 2.4);
  run<LDS | PERM, 2>("no MFMA: 2 perm + 2 ds_read_b128", it, 2.4);
  run<LDS, 2>("gathers only (uniform rows, no perm)", it, 2.4);
  run<MFMA | PERM, 2>("no LDS: 2 perm + 4 MFMA", it, 2.4);
  run<MFMA, 2>("MFMA only", it, 2.4);
  run<LDS | MFMA, 2>("no perm: 2 ds_read_b128 + 4 MFMA", it, 2.4);
  run_tokens<MFMA | PERM>("2-4 tokens: 10 perm + 2 ds_read + 4 MFMA, fenced", it, 2.4);
  run_tokens<MFMA>("2-4 tokens, transposition steps not fenced", it, 2.4);
  run_tokens<PERM>("2-4 tokens without the MFMAs (fenced)", it, 2.4);
  return 0;
}
