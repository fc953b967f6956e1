// Energy per wave-instruction and per index-wave on gfx950 (MI355X).  The one-token chain kernel runs at the 1400 W
// package power limit, so the figure of merit of a formulation is joules per index, not cycles.
// Every mode keeps all 256 CUs busy (16 waves per CU, registers / LDS only, no memory traffic except mode 60) for
// `seconds`; the binary itself samples hwmon power1_input / freq1_input of THIS GPU (matched by PCI address) every
// 10 ms and prints, per mode: rate, median package power, median shader clock, SIMD cycles per wave-instruction and
// (power - power of the s_nop mode) / rate = joules per wave-instruction at the clock the mode settles at.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -I vptq_amd/csrc tools/ubench_energy.hip -o tools/_build/ubench_energy -lpthread
//   tools/_build/ubench_energy [seconds per mode = 1.5] [mode list, default all]
// Single-instruction modes:
//   0 v_mfma_f32_4x4x4_16B_f16   10 the same with a one-hot A operand (x * e_j: what the kernels feed it)
//   1 v_mfma_f32_16x16x16_f16    2 v_mfma_f32_16x16x32_f16
//   4 v_perm_b32   5 v_dot2_f32_f16   6 v_pk_fma_f16   7 v_fma_mix_f32   8 v_pk_add_f16   15 v_pk_fma_f32
//   16 v_fma_f32   17 v_pk_mul_f16   18 v_and_b32   19 s_add_u32 (16 per wave-"instruction" slot)   9 s_nop (baseline)
//   11 ds_read_b128, conflict-free gather addresses (precomputed, rotating; results not used)
//   12 ds_read_b64 (same)   13 ds_read_b32 (same)   14 ds_bpermute_b32
// Inner loops of a one-token dequant-GEMV, per index-wave (one index = 8 + 8 halves for each of 64 lanes); all of them
// 2 address perms + 2 ds_read_b128 gathers per index, index words from a per-lane counter:
//   40 folded MFMA form of gemv_k256c: 4 v_mfma_4x4x4 + x-operand perms shared by 2 row subgroups
//   41 f16(c + r) first: 4 v_pk_add_f16 + 2 v_mfma_4x4x4
//   42 VALU, fp32 accumulate: 4 v_pk_add_f16 + 8 v_fma_mix_f32 (x' selected by op_sel, no operand perms)
//   43 VALU, packed f16 accumulate over the 8 columns of a lane's chunk, then widened: 4 v_pk_add_f16 + 4 v_pk_fma_f16
//      per index + 8 v_fma_mix_f32 per 8 indices
//   44 VALU, fp32 accumulate, no pre-add: 16 v_fma_mix_f32
//   45 the gathers alone (2 perms + 2 ds_read_b128)
//   46 as 42 with v_dot2_f32_f16 on (c, r) pairs interleaved by v_perm: 8 perms + 8 dot2  (c + r never rounded)
//   47 the reference's roundings r16(r16(r16(c + r) s) + b): 4 v_pk_add + 4 v_pk_mul + 4 v_pk_add + 2 v_mfma_4x4x4
// 60 HBM stream: every CU reads a 2 GiB buffer with 16-byte non-temporal loads (pJ per byte of the memory system)
#include <hip/hip_runtime.h>
#include <dirent.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <string>
#include <thread>
#include <vector>
#include "common.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
using namespace vptq;
typedef _Float16 h8_t __attribute__((ext_vector_type(8)));

// ---- hwmon sampling -----------------------------------------------------------------------------------------
static std::string g_hwmon;
static void find_hwmon() {
  char bus[64] = {0};
  if (hipDeviceGetPCIBusId(bus, sizeof bus, 0) != hipSuccess) return;
  for (char* p = bus; *p; ++p) *p = (char)tolower(*p);
  const std::string base = std::string("/sys/bus/pci/devices/") + bus + "/hwmon";
  DIR* d = opendir(base.c_str());
  if (!d) return;
  while (dirent* e = readdir(d))
    if (!strncmp(e->d_name, "hwmon", 5)) { g_hwmon = base + "/" + e->d_name; break; }
  closedir(d);
}
static double read_num(const std::string& p) {
  FILE* f = fopen(p.c_str(), "r");
  if (!f) return -1.0;
  double v = -1.0;
  if (fscanf(f, "%lf", &v) != 1) v = -1.0;
  fclose(f);
  return v;
}
struct Sampler {
  std::atomic<bool> stop{false};
  std::vector<double> w, mhz;
  std::thread th;
  void start() {
    th = std::thread([this] {
      while (!stop) {
        if (!g_hwmon.empty()) {
          const double p = read_num(g_hwmon + "/power1_input"), f = read_num(g_hwmon + "/freq1_input");
          if (p >= 0) w.push_back(p * 1e-6);
          if (f >= 0) mhz.push_back(f * 1e-6);
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(10));
      }
    });
  }
  void finish() { stop = true; th.join(); }
  // median of the samples after the first 30 % (the clock settles within a few hundred ms)
  static double med(std::vector<double> v) {
    if (v.empty()) return -1.0;
    v.erase(v.begin(), v.begin() + v.size() * 3 / 10);
    std::sort(v.begin(), v.end());
    return v[v.size() / 2];
  }
};

// ---- kernels ------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) u32x2 lds_u32x2_t;
typedef __attribute__((address_space(3))) uint32_t lds_u32_t;

template <int MODE>
__global__ __launch_bounds__(1024) void spin(float* out, int iters, uint32_t seed) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // LDS: 64 KiB "image" of half-precision values around 0.01 ... 1 with random mantissas, + 4 KiB of activations
  for (int i = tid; i < 16384 + 1024; i += 1024) {
    const uint32_t h = (uint32_t)i * 2654435761u + seed;
    ((uint32_t*)smem)[i] = (h & 0x03ff03ffu) | 0x2c002c00u | ((h >> 3) & 0x80008000u);
  }
  __syncthreads();
  uint32_t r = (tid * 2654435761u + seed) ^ (blockIdx.x * 40503u);
  f32x4 acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  auto gen = [&]() { r = r * 1664525u + 1013904223u; return (r & 0x03ff03ffu) | 0x2c002c00u | (r & 0x80008000u); };
  uint32_t a0 = gen(), a1 = gen(), b0 = gen(), b1 = gen(), b2 = gen(), b3 = gen();
  const int j = lane & 3;
  const uint32_t oh0 = j == 0 ? (a0 & 0xffffu) : j == 1 ? (a0 << 16) : 0u, oh1 = j == 2 ? (a0 & 0xffffu) : j == 3 ? (a0 << 16) : 0u;
  float fa = 1.0f + lane * 0.001f, fb = 0.5f + lane * 0.002f;
  uint32_t pk = a0, pk2 = a1;
  float res = 0.f;

  if constexpr (MODE < 40) {
    // conflict-free gather addresses: lane l reads unit (l & 15) of a random image row; 16 of them, precomputed
    uint32_t ga[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) { r = r * 1664525u + 1013904223u; ga[u] = (((r >> 10) & 255u) << 8) | ((uint32_t)(lane & 15) << 4); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        if constexpr (MODE == 0 || MODE == 10) {
          const uint32_t x0 = MODE == 10 ? oh0 : a0, x1 = MODE == 10 ? oh1 : a1;
          acc[u & 3] = F16::mfma4(u32x2{x0, x1}, u32x2{b0 + (uint32_t)u, b1}, acc[u & 3]);
        } else if constexpr (MODE == 1) {
          const h4_t A = __builtin_bit_cast(h4_t, u32x2{a0, a1});
          const h4_t B = __builtin_bit_cast(h4_t, u32x2{b0 + (uint32_t)u, b1});
          acc[u & 3] = __builtin_amdgcn_mfma_f32_16x16x16f16(A, B, acc[u & 3], 0, 0, 0);
        } else if constexpr (MODE == 2) {
          const u32x4 Au = {a0, a1, b2, b3}, Bu = {b0 + (uint32_t)u, b1, a1, a0};
          acc[u & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8_t, Au), __builtin_bit_cast(h8_t, Bu), acc[u & 3], 0, 0, 0);
        } else if constexpr (MODE == 11) {
          u32x4 q;
          asm volatile("ds_read_b128 %0, %1" : "=v"(q) : "v"(ga[u]) : "memory");
          if (u == 15) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if constexpr (MODE == 12) {
          u32x2 q;
          asm volatile("ds_read_b64 %0, %1" : "=v"(q) : "v"(ga[u]) : "memory");
          if (u == 15) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if constexpr (MODE == 13) {
          uint32_t q;
          asm volatile("ds_read_b32 %0, %1" : "=v"(q) : "v"(ga[u]) : "memory");
          if (u == 15) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if constexpr (MODE == 14) {
          uint32_t q;
          asm volatile("ds_bpermute_b32 %0, %1, %2" : "=v"(q) : "v"(ga[u]), "v"(b0) : "memory");
          if (u == 15) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if constexpr (MODE == 4) {
          pk = __builtin_amdgcn_perm(pk, b0 + u, 0x05040100u ^ (uint32_t)(u & 1) * 0x02020202u);
        } else if constexpr (MODE == 5) {
          fa = F16::dot2(a0 + u, b0, fa);
        } else if constexpr (MODE == 6) {
          asm("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(pk) : "v"(a0 + (uint32_t)u), "v"(b0));
        } else if constexpr (MODE == 7) {
          asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,0]" : "+v"(fa) : "v"(a0 + (uint32_t)u), "v"(b0));
        } else if constexpr (MODE == 8) {
          pk = F16::add2(pk, b0 + (uint32_t)u);
        } else if constexpr (MODE == 15) {
          typedef float f2_t __attribute__((ext_vector_type(2)));
          f2_t p = {fa, fb};
          const f2_t m = {__uint_as_float(0x3f7ff000u + (uint32_t)u), 0.99993f}, c = {__uint_as_float(a0 & 0x3fffffffu), 1e-3f};
          asm("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p) : "v"(m), "v"(c));
          fa = p[0]; fb = p[1];
        } else if constexpr (MODE == 16) {
          asm("v_fma_f32 %0, %0, %1, %2" : "+v"(fa) : "v"(__uint_as_float(0x3f7ff000u + (uint32_t)u)), "v"(__uint_as_float(a0 & 0x3fffffffu)));
        } else if constexpr (MODE == 17) {
          pk = F16::mul2(pk, (b0 & 0x03ff03ffu) | 0x3c003c00u);
        } else if constexpr (MODE == 18) {
          asm("v_and_b32 %0, %0, %1" : "+v"(pk) : "v"(b0 | (0xffff0000u + (uint32_t)u)));
        } else if constexpr (MODE == 19) {
          uint32_t s = (uint32_t)__builtin_amdgcn_readfirstlane((int)pk2);
          asm volatile("s_add_u32 %0, %0, 0x1234567\n\ts_add_u32 %0, %0, 0x7654321\n\ts_add_u32 %0, %0, 0x1234567\n\ts_add_u32 %0, %0, 0x7654321\n\t"
                       "s_add_u32 %0, %0, 0x1234567\n\ts_add_u32 %0, %0, 0x7654321\n\ts_add_u32 %0, %0, 0x1234567\n\ts_add_u32 %0, %0, 0x7654321\n\t"
                       "s_add_u32 %0, %0, 0x1234567\n\ts_add_u32 %0, %0, 0x7654321\n\ts_add_u32 %0, %0, 0x1234567\n\ts_add_u32 %0, %0, 0x7654321\n\t"
                       "s_add_u32 %0, %0, 0x1234567\n\ts_add_u32 %0, %0, 0x7654321\n\ts_add_u32 %0, %0, 0x1234567\n\ts_add_u32 %0, %0, 0x7654321"
                       : "+s"(s) :: "scc");
          if (u == 15) pk2 = s;
        } else {
          asm volatile("s_nop 3");
        }
      }
    }
    res = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + fa + fb + __uint_as_float((pk ^ pk2) & 0x3fffffffu);
  } else {
    // ---- inner loops: lane = (column chunk of 8, vector-row j); per step 8 columns x 2 row subgroups = 16 index-waves
    const uint32_t hi = (lane >> 3) & 1u;
    const uint32_t baseA = ((hi << 3) | (lane & 7u)) << 4;
    const uint32_t baseB = (((hi ^ 1u) << 3) | (lane & 7u)) << 4;
    const uint32_t selGA[2] = {0x0c0c0400u | (hi << 8), 0x0c0c0600u | (hi << 8)};
    const uint32_t selGB[2] = {0x0c0c0400u | ((hi ^ 1u) << 8), 0x0c0c0600u | ((hi ^ 1u) << 8)};
    const uint32_t selA[2] = {j == 0 ? 0x0c0c0504u : j == 1 ? 0x05040c0cu : 0x0c0c0c0cu, j == 0 ? 0x0c0c0706u : j == 1 ? 0x07060c0cu : 0x0c0c0c0cu};
    const uint32_t selB[2] = {j == 2 ? 0x0c0c0504u : j == 3 ? 0x05040c0cu : 0x0c0c0c0cu, j == 2 ? 0x0c0c0706u : j == 3 ? 0x07060c0cu : 0x0c0c0c0cu};
    f32x4 am[2][2] = {{{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}};
    float av[2][8];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int t = 0; t < 8; ++t) av[q][t] = 0.f;
    u32x4 words[2] = {u32x4{gen(), gen(), gen(), gen()}, u32x4{gen(), gen(), gen(), gen()}};
    constexpr int AH = 2, NB = AH + 1;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int k = 0; k < 4; ++k) words[q][k] += 0x9e3779b9u + (uint32_t)k * 0x01010101u;   // (one VALU op per 2 indices)
      const u32x4 xq = lds_load16(65536u + (uint32_t)(wave * 16 + (lane >> 2)) * 16u);
      u32x4 cv[NB], rv[NB];
      auto gather = [&](int t) {
        const int u = t >> 1, q = t & 1;
        const uint32_t w = words[q][u >> 1];
        cv[t % NB] = lds_load16(__builtin_amdgcn_perm(w, baseA, selGA[u & 1]));
        rv[t % NB] = lds_load16(__builtin_amdgcn_perm(w, baseB, selGB[u & 1]));
      };
#pragma unroll
      for (int t = 0; t < AH; ++t) gather(t);
      u32x2 xo = u32x2{0u, 0u};
      uint32_t a16[2][4];   // mode 43: packed f16 sums of the chunk
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        __builtin_amdgcn_sched_barrier(0);
        if (t + AH < 16) gather(t + AH);
        __builtin_amdgcn_sched_barrier(0);
        const int u = t >> 1, q = t & 1;
        const u32x4 c = cv[t % NB], rr = rv[t % NB];
        const uint32_t xp = xq[u >> 1];   // x' of column u = half (u & 1)
        if constexpr (MODE == 40) {
          if (q == 0) xo = u32x2{__builtin_amdgcn_perm(xp, 0u, selA[u & 1]), __builtin_amdgcn_perm(xp, 0u, selB[u & 1])};
          am[q][0] = F16::mfma4(xo, u32x2{c[0], c[1]}, am[q][0]);
          am[q][1] = F16::mfma4(xo, u32x2{c[2], c[3]}, am[q][1]);
          am[q][0] = F16::mfma4(xo, u32x2{rr[0], rr[1]}, am[q][0]);
          am[q][1] = F16::mfma4(xo, u32x2{rr[2], rr[3]}, am[q][1]);
        } else if constexpr (MODE == 41) {
          if (q == 0) xo = u32x2{__builtin_amdgcn_perm(xp, 0u, selA[u & 1]), __builtin_amdgcn_perm(xp, 0u, selB[u & 1])};
          const u32x4 w = u32x4{F16::add2(c[0], rr[0]), F16::add2(c[1], rr[1]), F16::add2(c[2], rr[2]), F16::add2(c[3], rr[3])};
          am[q][0] = F16::mfma4(xo, u32x2{w[0], w[1]}, am[q][0]);
          am[q][1] = F16::mfma4(xo, u32x2{w[2], w[3]}, am[q][1]);
        } else if constexpr (MODE == 42) {
          // (stage-major: the four adds, then the eight multiply-adds; the empty statement keeps the compiler from
          // re-pairing every add with its two users, which costs a wait state each)
          uint32_t w[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) w[k] = F16::add2(c[k], rr[k]);
          asm volatile("" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]));
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            av[q][2 * k] = F16::fma_lo_h(w[k], xp, u & 1, av[q][2 * k]);
            av[q][2 * k + 1] = F16::fma_hi_h(w[k], xp, u & 1, av[q][2 * k + 1]);
          }
        } else if constexpr (MODE == 43) {
          // (stage-major: the four adds, then the four multiply-adds - dependent packed ops never back to back)
          uint32_t w[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) w[k] = F16::add2(c[k], rr[k]);
          asm volatile("" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]));
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            if (u == 0) {
              a16[q][k] = F16::mul2_bcast(w[k], xp, 0);
            } else if (u & 1) {
              asm("v_pk_fma_f16 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(a16[q][k]) : "v"(w[k]), "v"(xp));
            } else {
              asm("v_pk_fma_f16 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "+v"(a16[q][k]) : "v"(w[k]), "v"(xp));
            }
          }
          if (u == 7) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              asm("v_fma_mix_f32 %0, %1, 1.0, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(av[q][2 * k]) : "v"(a16[q][k]));
              asm("v_fma_mix_f32 %0, %1, 1.0, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(av[q][2 * k + 1]) : "v"(a16[q][k]));
            }
          }
        } else if constexpr (MODE == 47) {
          // the reference's three roundings per weight, r16(r16(r16(c + r) s) + b), then 2 MFMAs with the raw x:
          // 4 v_pk_add_f16 + 4 v_pk_mul_f16 + 4 v_pk_add_f16 (scale / bias of the column by op_sel) + 2 v_mfma_4x4x4.
          // Both row subgroups of a column together, stage-major over 8 independent chains (dependent packed ops
          // back to back cost wait states).
          if (q == 0) {
            xo = u32x2{__builtin_amdgcn_perm(xp, 0u, selA[u & 1]), __builtin_amdgcn_perm(xp, 0u, selB[u & 1])};
            const uint32_t sp = xq[(u >> 1) ^ 1], bp = xq[(u >> 1) ^ 2];   // (stand-ins for the staged scale / bias pairs)
            const u32x4 c1 = cv[(t + 1) % NB], r1 = rv[(t + 1) % NB];
            uint32_t w[8];
#pragma unroll
            for (int k = 0; k < 4; ++k) { w[k] = F16::add2(c[k], rr[k]); w[4 + k] = F16::add2(c1[k], r1[k]); }
            asm volatile("" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7]));
#pragma unroll
            for (int k = 0; k < 8; ++k) w[k] = F16::mul2_bcast(w[k], sp, u & 1);
            asm volatile("" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7]));
#pragma unroll
            for (int k = 0; k < 8; ++k) w[k] = F16::add2_bcast(w[k], bp, u & 1);
            am[0][0] = F16::mfma4(xo, u32x2{w[0], w[1]}, am[0][0]);
            am[0][1] = F16::mfma4(xo, u32x2{w[2], w[3]}, am[0][1]);
            am[1][0] = F16::mfma4(xo, u32x2{w[4], w[5]}, am[1][0]);
            am[1][1] = F16::mfma4(xo, u32x2{w[6], w[7]}, am[1][1]);
          }
        } else if constexpr (MODE == 44) {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            av[q][2 * k] = F16::fma_lo_h(c[k], xp, u & 1, av[q][2 * k]);
            av[q][2 * k + 1] = F16::fma_hi_h(c[k], xp, u & 1, av[q][2 * k + 1]);
            av[q][2 * k] = F16::fma_lo_h(rr[k], xp, u & 1, av[q][2 * k]);
            av[q][2 * k + 1] = F16::fma_hi_h(rr[k], xp, u & 1, av[q][2 * k + 1]);
          }
        } else if constexpr (MODE == 46) {
          const uint32_t xx = __builtin_amdgcn_perm(xp, xp, (u & 1) ? 0x07060706u : 0x05040504u);   // (x', x'); hoistable per column
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const uint32_t lo = __builtin_amdgcn_perm(rr[k], c[k], 0x05040100u);   // (c.lo, r.lo)
            const uint32_t hi2 = __builtin_amdgcn_perm(rr[k], c[k], 0x07060302u);  // (c.hi, r.hi)
            av[q][2 * k] = F16::dot2(lo, xx, av[q][2 * k]);
            av[q][2 * k + 1] = F16::dot2(hi2, xx, av[q][2 * k + 1]);
          }
        } else {
          asm volatile("" :: "v"(c), "v"(rr));
        }
      }
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      res += am[q][0][0] + am[q][1][1] + am[q][0][2] + am[q][1][3];
#pragma unroll
      for (int t = 0; t < 8; ++t) res += av[q][t];
    }
  }
  if (res == 1234.5678f) out[tid] = res;
}

// HBM stream: persistent, 16 waves per CU, every lane 4 x 16 bytes in flight
__global__ __launch_bounds__(1024) void stream_kernel(const u32x4* __restrict__ buf, size_t n16, float* out) {
  const size_t stride = (size_t)gridDim.x * 1024 * 4;
  u32x4 s = {0u, 0u, 0u, 0u};
  for (size_t i = (size_t)blockIdx.x * 1024 * 4 + threadIdx.x; i + 3 * 1024 < n16; i += stride) {
    const u32x4 a = __builtin_nontemporal_load(buf + i), b = __builtin_nontemporal_load(buf + i + 1024);
    const u32x4 c = __builtin_nontemporal_load(buf + i + 2048), d = __builtin_nontemporal_load(buf + i + 3072);
    s ^= a ^ b ^ c ^ d;
  }
  if ((s[0] ^ s[1] ^ s[2] ^ s[3]) == 0x12345u) out[threadIdx.x] = 1.f;
}

static double g_nop_w = -1.0;
struct Result { double rate, w, mhz; };

template <int MODE>
static Result run(float* out, double seconds, const char* name, double per_iter, const char* unit) {
  CK(hipFuncSetAttribute((const void*)spin<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024));
  const int iters = MODE >= 40 ? 2000 : 20000;
  hipLaunchKernelGGL((spin<MODE>), dim3(256), dim3(1024), 72 * 1024, 0, out, 100, 1u);
  CK(hipDeviceSynchronize());
  Sampler sm; sm.start();
  auto t0 = std::chrono::steady_clock::now();
  long long launches = 0;
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
    for (int k = 0; k < 4; ++k) hipLaunchKernelGGL((spin<MODE>), dim3(256), dim3(1024), 72 * 1024, 0, out, iters, (uint32_t)launches + k);
    launches += 4;
    CK(hipDeviceSynchronize());
  }
  const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  sm.finish();
  const double n = (double)launches * iters * per_iter * 256.0 * 16.0;   // wave-level units of the measured kind
  Result R{n / dt, Sampler::med(sm.w), Sampler::med(sm.mhz)};
  if (MODE == 9) g_nop_w = R.w;
  const double cyc = R.mhz > 0 ? R.mhz * 1e6 * 1024.0 / R.rate : -1.0;   // SIMD cycles per unit (1024 SIMDs)
  const double nj = (g_nop_w > 0 && R.w > 0) ? (R.w - g_nop_w) / R.rate * 1e9 : -1.0;
  printf("mode %2d %-58s %.3e %s/s | %6.0f W | %5.0f MHz | %6.2f SIMD cycles per %s | %6.2f nJ per %s above the s_nop mode\n",
         MODE, name, R.rate, unit, R.w, R.mhz, cyc, unit, nj, unit);
  fflush(stdout);
  return R;
}

static void run_stream(float* out, double seconds) {
  const size_t bytes = (size_t)2 << 30;
  u32x4* buf; CK(hipMalloc(&buf, bytes)); CK(hipMemset(buf, 0x5a, bytes));
  hipLaunchKernelGGL(stream_kernel, dim3(256), dim3(1024), 0, 0, buf, bytes / 16, out);
  CK(hipDeviceSynchronize());
  Sampler sm; sm.start();
  auto t0 = std::chrono::steady_clock::now();
  long long passes = 0;
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
    for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(stream_kernel, dim3(256), dim3(1024), 0, 0, buf, bytes / 16, out);
    passes += 4;
    CK(hipDeviceSynchronize());
  }
  const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  sm.finish();
  const double rate = (double)passes * bytes / dt, w = Sampler::med(sm.w), mhz = Sampler::med(sm.mhz);
  printf("mode 60 %-58s %.3e B/s | %6.0f W | %5.0f MHz | %6.1f pJ per byte above the s_nop mode\n",
         "HBM stream, 2 GiB buffer, 16-byte nt loads", rate, w, mhz, g_nop_w > 0 ? (w - g_nop_w) / rate * 1e12 : -1.0);
  CK(hipFree(buf));
}

int main(int argc, char** argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 1.5;
  std::vector<int> want;
  for (int i = 2; i < argc; ++i) want.push_back(atoi(argv[i]));
  auto on = [&](int m) { return want.empty() || std::find(want.begin(), want.end(), m) != want.end(); };
  find_hwmon();
  printf("hwmon: %s   power cap %.0f W\n", g_hwmon.empty() ? "(none)" : g_hwmon.c_str(), g_hwmon.empty() ? -1.0 : read_num(g_hwmon + "/power1_cap") * 1e-6);
  float* out; CK(hipMalloc(&out, 4096 * 4));
  {
    Sampler sm; sm.start();
    std::this_thread::sleep_for(std::chrono::milliseconds(800));
    sm.finish();
    printf("idle (no kernel): %.0f W, %.0f MHz\n", Sampler::med(sm.w), Sampler::med(sm.mhz));
  }
  const char* wi = "wave-instr";
  run<9>(out, seconds, "s_nop 3 (16 resident waves per CU doing nothing)", 16, wi);   // always: the baseline
  if (on(0)) run<0>(out, seconds, "v_mfma_f32_4x4x4_16B_f16", 16, wi);
  if (on(10)) run<10>(out, seconds, "v_mfma_f32_4x4x4_16B_f16, one-hot A operand", 16, wi);
  if (on(1)) run<1>(out, seconds, "v_mfma_f32_16x16x16_f16", 16, wi);
  if (on(2)) run<2>(out, seconds, "v_mfma_f32_16x16x32_f16", 16, wi);
  if (on(4)) run<4>(out, seconds, "v_perm_b32", 16, wi);
  if (on(5)) run<5>(out, seconds, "v_dot2_f32_f16", 16, wi);
  if (on(6)) run<6>(out, seconds, "v_pk_fma_f16", 16, wi);
  if (on(7)) run<7>(out, seconds, "v_fma_mix_f32", 16, wi);
  if (on(8)) run<8>(out, seconds, "v_pk_add_f16", 16, wi);
  if (on(15)) run<15>(out, seconds, "v_pk_fma_f32", 16, wi);
  if (on(16)) run<16>(out, seconds, "v_fma_f32", 16, wi);
  if (on(17)) run<17>(out, seconds, "v_pk_mul_f16", 16, wi);
  if (on(18)) run<18>(out, seconds, "v_and_b32", 16, wi);
  if (on(19)) run<19>(out, seconds, "s_add_u32 (counted per scalar instruction)", 256, "scalar-instr");
  if (on(11)) run<11>(out, seconds, "ds_read_b128, conflict-free gather, result unused", 16, wi);
  if (on(12)) run<12>(out, seconds, "ds_read_b64, same addresses", 16, wi);
  if (on(13)) run<13>(out, seconds, "ds_read_b32, same addresses", 16, wi);
  if (on(14)) run<14>(out, seconds, "ds_bpermute_b32", 16, wi);
  const char* iw = "index-wave";
  if (on(45)) run<45>(out, seconds, "loop: gathers only (2 v_perm + 2 ds_read_b128 per index)", 16, iw);
  if (on(40)) run<40>(out, seconds, "loop: folded MFMA form (4 v_mfma_4x4x4; gemv_k256c today)", 16, iw);
  if (on(41)) run<41>(out, seconds, "loop: f16(c+r) first, 4 v_pk_add_f16 + 2 v_mfma_4x4x4", 16, iw);
  if (on(42)) run<42>(out, seconds, "loop: VALU fp32 accumulate, 4 v_pk_add_f16 + 8 v_fma_mix_f32", 16, iw);
  if (on(43)) run<43>(out, seconds, "loop: VALU f16 chunk sums, 4 v_pk_add + 4 v_pk_fma_f16 + 1 widen", 16, iw);
  if (on(44)) run<44>(out, seconds, "loop: VALU fp32 accumulate, no pre-add, 16 v_fma_mix_f32", 16, iw);
  if (on(46)) run<46>(out, seconds, "loop: 8 v_perm + 8 v_dot2_f32_f16 on (c, r) pairs", 16, iw);
  if (on(47)) run<47>(out, seconds, "loop: reference roundings, 12 packed VALU + 2 v_mfma_4x4x4", 16, iw);
  if (on(60)) run_stream(out, seconds);
  return 0;
}
