// Energy per instruction on gfx950 (MI355X): the one-token chain kernel runs at the 1400 W package power limit
// (DESIGN.md 4.9), so the figure of merit of a formulation is joules per index, not cycles.  Each mode keeps all 256 CUs
// busy with ONE kind of instruction (16 waves per CU, registers only / LDS only, no memory traffic) for `seconds`;
// run it under the power probe and divide:
//   python tools/power_probe.py --exe "tools/_build/ubench_energy <mode> 3"
//   (package power - idle power) / (wave-instructions per second) = joules per wave-instruction at the clock it settles at
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_energy.hip -o tools/_build/ubench_energy
// modes: 0 v_mfma_f32_4x4x4_16B_f16   1 v_mfma_f32_16x16x16_f16   2 v_mfma_f32_16x16x32_f16   3 ds_read_b128 (conflict-free gather)
//        4 v_perm_b32   5 v_dot2_f32_f16   6 v_pk_fma_f16   7 v_fma_mix_f32   8 v_pk_add_f16   9 s_nop (idle waves)
//        10 4x4x4 MFMA with a one-hot A operand (x * e_j: what the kernel feeds it)
// Modes 0, 10 and 3 were run once (profiles/r03/ubench_energy_first.txt); the others are next round's first measurement.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(1024) void spin(float* out, int iters, uint32_t seed) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 16384; i += 1024) ((uint32_t*)smem)[i] = 0x3c003c00u + (uint32_t)i * 2654435761u;
  __syncthreads();
  uint32_t r = (tid * 2654435761u + seed) ^ (blockIdx.x * 40503u);
  f32x4 acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  // operands with realistic bit activity: half-precision values around 0.01 ... 1 from a per-lane generator
  auto gen = [&]() { r = r * 1664525u + 1013904223u; return (r & 0x03ff03ffu) | 0x2c002c00u | (r & 0x80008000u); };
  uint32_t a0 = gen(), a1 = gen(), b0 = gen(), b1 = gen(), b2 = gen(), b3 = gen();
  const int j = lane & 3;
  const uint32_t oh0 = j == 0 ? (a0 & 0xffffu) : j == 1 ? (a0 << 16) : 0u, oh1 = j == 2 ? (a0 & 0xffffu) : j == 3 ? (a0 << 16) : 0u;
  float fa = 1.0f + lane * 0.001f;
  uint32_t pk = a0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if constexpr (MODE == 0 || MODE == 10) {
        const uint32_t x0 = MODE == 10 ? oh0 : a0, x1 = MODE == 10 ? oh1 : a1;
        const h4_t A = __builtin_bit_cast(h4_t, (uint64_t)x0 | ((uint64_t)x1 << 32));
        const h4_t B = __builtin_bit_cast(h4_t, (uint64_t)(b0 + u) | ((uint64_t)b1 << 32));
        acc[u & 3] = __builtin_amdgcn_mfma_f32_4x4x4f16(A, B, acc[u & 3], 0, 0, 0);
      } else if constexpr (MODE == 1) {
        const h4_t A = __builtin_bit_cast(h4_t, (uint64_t)a0 | ((uint64_t)a1 << 32));
        const h4_t B = __builtin_bit_cast(h4_t, (uint64_t)(b0 + u) | ((uint64_t)b1 << 32));
        acc[u & 3] = __builtin_amdgcn_mfma_f32_16x16x16f16(A, B, acc[u & 3], 0, 0, 0);
      } else if constexpr (MODE == 2) {
        const u32x4 Au = {a0, a1, b2, b3}, Bu = {b0 + (uint32_t)u, b1, a1, a0};
        acc[u & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8_t, Au), __builtin_bit_cast(h8_t, Bu), acc[u & 3], 0, 0, 0);
      } else if constexpr (MODE == 3) {
        r = r * 1664525u + 1013904223u;
        const uint32_t a = (((r >> 10) & 255u) << 8) | ((uint32_t)(lane & 15) << 4);
        typedef __attribute__((address_space(3))) f32x4 lds_f4;
        const f32x4 q = *(const lds_f4*)(uintptr_t)a;
        acc[u & 3] += q;
      } else if constexpr (MODE == 4) {
        pk = __builtin_amdgcn_perm(pk, b0 + u, 0x05040100u ^ (uint32_t)(u & 1) * 0x02020202u);
      } else if constexpr (MODE == 5) {
        fa = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2_t, a0 + u), __builtin_bit_cast(h2_t, b0), fa, false);
      } else if constexpr (MODE == 6) {
        h2_t p = __builtin_bit_cast(h2_t, pk);
        p = __builtin_elementwise_fma(__builtin_bit_cast(h2_t, a0 + u), __builtin_bit_cast(h2_t, b0), p);
        pk = __builtin_bit_cast(uint32_t, p);
      } else if constexpr (MODE == 7) {
        const uint32_t ea = a0 + u;
        asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,0]" : "+v"(fa) : "v"(ea), "v"(b0));
      } else if constexpr (MODE == 8) {
        h2_t p = __builtin_bit_cast(h2_t, pk) + __builtin_bit_cast(h2_t, b0 + u);
        pk = __builtin_bit_cast(uint32_t, p);
      } else {
        asm volatile("s_nop 3");
      }
    }
  }
  const float res = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + fa + __uint_as_float(pk & 0x3fffffffu);
  if (res == 1234.5678f) out[tid] = res;
}

template <int MODE>
static void run(float* out, double seconds, const char* name) {
  CK(hipFuncSetAttribute((const void*)spin<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  const int iters = 20000;
  hipLaunchKernelGGL((spin<MODE>), dim3(256), dim3(1024), 65536, 0, out, 100, 1u);
  CK(hipDeviceSynchronize());
  auto t0 = std::chrono::steady_clock::now();
  long long launches = 0;
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
    for (int k = 0; k < 4; ++k) hipLaunchKernelGGL((spin<MODE>), dim3(256), dim3(1024), 65536, 0, out, iters, (uint32_t)launches);
    launches += 4;
    CK(hipDeviceSynchronize());
  }
  const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  const double winstr = (double)launches * iters * 16.0 * 256.0 * 16.0;   // wave-instructions of the measured kind
  printf("mode %-44s %.3e wave-instructions / s over %.1f s (%.2f per CU and ns)\n", name, winstr / dt, dt, winstr / dt / 256.0 / 1e9);
}

int main(int argc, char** argv) {
  const int mode = argc > 1 ? atoi(argv[1]) : 0;
  const double seconds = argc > 2 ? atof(argv[2]) : 2.0;
  float* out; CK(hipMalloc(&out, 4096 * 4));
  switch (mode) {
    case 0: run<0>(out, seconds, "0 v_mfma_f32_4x4x4_16B_f16"); break;
    case 1: run<1>(out, seconds, "1 v_mfma_f32_16x16x16_f16"); break;
    case 2: run<2>(out, seconds, "2 v_mfma_f32_16x16x32_f16"); break;
    case 3: run<3>(out, seconds, "3 ds_read_b128 gather"); break;
    case 4: run<4>(out, seconds, "4 v_perm_b32"); break;
    case 5: run<5>(out, seconds, "5 v_dot2_f32_f16"); break;
    case 6: run<6>(out, seconds, "6 v_pk_fma_f16"); break;
    case 7: run<7>(out, seconds, "7 v_fma_mix_f32"); break;
    case 8: run<8>(out, seconds, "8 v_pk_add_f16"); break;
    case 10: run<10>(out, seconds, "10 v_mfma_f32_4x4x4 with a one-hot A operand"); break;
    default: run<9>(out, seconds, "9 s_nop (idle waves)"); break;
  }
  return 0;
}
