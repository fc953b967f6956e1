// Instruction-throughput micro-benchmark for gfx950 (MI355X).  Build & run on the GPU box:
//   hipcc --offload-arch=gfx950 -O2 tools/ubench.hip -o /tmp/ubench && /tmp/ubench
// Reports ns per wave-instruction per SIMD for several VALU / MFMA / LDS ops at 1, 2, 4
// waves per SIMD.  Used to decide which arithmetic formulation the GEMV kernel can afford.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

constexpr int UNROLL = 16;

#define REP16(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15)

template <int OP>
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned seed) {
  // 16 independent accumulator registers
  unsigned r[UNROLL];
  float f[UNROLL];
  const unsigned t = threadIdx.x + blockIdx.x * 256;
  for (int i = 0; i < UNROLL; ++i) { r[i] = (t * 2654435761u + i * 40503u + seed) & 0x3bff3bffu; f[i] = 1.0f + (float)(t & 15) * 0.001f + i; }
  unsigned a = 0x3c003c00u ^ (seed & 1), b = 0x38003800u | (seed & 2);
  float fa = 1.0001f, fb = 0.9999f;
  float fa2[2] = {1.0001f, 0.9998f}, fb2[2] = {1e-3f, 2e-3f};
  f4 acc4[4]; for (int i = 0; i < 4; ++i) acc4[i] = f4{0, 0, 0, 0};
  __shared__ __attribute__((aligned(16))) unsigned lds[4096 * 4];
  if (OP >= 20) { for (int i = threadIdx.x; i < 4096 * 4; i += 256) lds[i] = i * seed; __syncthreads(); }
  unsigned laddr = ((threadIdx.x & 15) << 4) | (((t * 7 + seed) & 255) << 8);  // conflict-free slots
  unsigned laddr_rand = ((t * 2654435761u + seed) >> 8) & 0xfff0u;
  for (int it = 0; it < iters; ++it) {
    if (OP == 0) {
#define S(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(fa), "v"(fb));
      REP16(S)
#undef S
    } else if (OP == 1) {
#define S(i) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(r[i]) : "v"(a));
      REP16(S)
#undef S
    } else if (OP == 2) {
#define S(i) asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(r[i]) : "v"(a));
      REP16(S)
#undef S
    } else if (OP == 3) {
#define S(i) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
      REP16(S)
#undef S
    } else if (OP == 4) {
#define S(i) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,1,0]" : "+v"(f[i]) : "v"(a), "v"(b));
      REP16(S)
#undef S
    } else if (OP == 5) {
#define S(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
      REP16(S)
#undef S
    } else if (OP == 6) {
#define S(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
      REP16(S)
#undef S
    } else if (OP == 7) {
#define S(i) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(f[i]) : "v"(a), "v"(b));
      REP16(S)
#undef S
    } else if (OP == 8) {
#define S(i) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(f[i]) : "v"(r[i]));
      REP16(S)
#undef S
    } else if (OP == 9) {
#define S(i) asm volatile("v_add_f16 %0, %0, %1" : "+v"(r[i]) : "v"(a));
      REP16(S)
#undef S
    } else if (OP == 13) {
      // 16 x v_pk_fma_f32 (2 fp32 FMAs per lane per instruction)
      for (int j = 0; j < 16; j += 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(*(double*)&f[j]) : "v"(*(double*)&fa2), "v"(*(double*)&fb2));
      for (int j = 0; j < 16; j += 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(*(double*)&f[j]) : "v"(*(double*)&fb2), "v"(*(double*)&fa2));
    } else if (OP == 10) {  // 16 x mfma 4x4x4 (16 blocks) f16
      h4 av = __builtin_bit_cast(h4, (unsigned long long)a | ((unsigned long long)b << 32));
      h4 bv = __builtin_bit_cast(h4, (unsigned long long)b | ((unsigned long long)a << 32));
      for (int j = 0; j < 4; ++j) {
        acc4[0] = __builtin_amdgcn_mfma_f32_4x4x4f16(av, bv, acc4[0], 0, 0, 0);
        acc4[1] = __builtin_amdgcn_mfma_f32_4x4x4f16(av, bv, acc4[1], 0, 0, 0);
        acc4[2] = __builtin_amdgcn_mfma_f32_4x4x4f16(av, bv, acc4[2], 0, 0, 0);
        acc4[3] = __builtin_amdgcn_mfma_f32_4x4x4f16(av, bv, acc4[3], 0, 0, 0);
      }
    } else if (OP == 11) {  // 16 x mfma 16x16x32 f16
      h8 av, bv;
      for (int j = 0; j < 8; ++j) { av[j] = (_Float16)(1.0f + j); bv[j] = (_Float16)(0.5f); }
      for (int j = 0; j < 4; ++j) {
        acc4[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, acc4[0], 0, 0, 0);
        acc4[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, acc4[1], 0, 0, 0);
        acc4[2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, acc4[2], 0, 0, 0);
        acc4[3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, acc4[3], 0, 0, 0);
      }
    } else if (OP == 12) {  // 8 mfma 4x4x4 interleaved with 16 pk_add (co-issue test)
      h4 av = __builtin_bit_cast(h4, (unsigned long long)a | ((unsigned long long)b << 32));
      h4 bv = __builtin_bit_cast(h4, (unsigned long long)b | ((unsigned long long)a << 32));
#define S(i) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(r[i]) : "v"(a)); \
      if ((i & 1) == 0) acc4[(i >> 1) & 3] = __builtin_amdgcn_mfma_f32_4x4x4f16(av, bv, acc4[(i >> 1) & 3], 0, 0, 0);
      REP16(S)
#undef S
    } else if (OP == 20) {  // 16 x ds_read_b128, bank-partitioned addresses (conflict free)
      typedef __attribute__((address_space(3))) const u4 lds4;
      for (int j = 0; j < 16; ++j) {
        unsigned ad = (laddr + j * 4096 + (r[j] & 0)) & 0xfff0u;
        u4 v = *(lds4*)(unsigned long)ad;
        r[j] ^= v.x ^ v.y ^ v.z ^ v.w;
      }
    } else if (OP == 21) {  // 16 x ds_read_b128, random 16-byte slots
      typedef __attribute__((address_space(3))) const u4 lds4;
      for (int j = 0; j < 16; ++j) {
        unsigned ad = (laddr_rand * (j * 2 + 1) + j * 1232) & 0xfff0u;
        u4 v = *(lds4*)(unsigned long)ad;
        r[j] ^= v.x ^ v.y ^ v.z ^ v.w;
      }
    } else if (OP == 22) {  // 16 x ds_write_b128 rotating slots
      typedef __attribute__((address_space(3))) u4 lds4w;
      for (int j = 0; j < 16; ++j) {
        unsigned ad = (((threadIdx.x & 255) << 8) | (((j + threadIdx.x) & 15) << 4)) & 0xfff0u;
        *(lds4w*)(unsigned long)ad = u4{r[j], r[0], r[1], r[2]};
      }
    }
  }
  float s = 0;
  for (int i = 0; i < UNROLL; ++i) s += f[i] + (float)(r[i] & 0xff);
  for (int i = 0; i < 4; ++i) s += acc4[i][0] + acc4[i][1] + acc4[i][2] + acc4[i][3];
  if (s == 12345.678f) out[t] = s;
}

template <int OP>
double run(const char* name, int waves_per_simd, int iters) {
  // 256 CUs x 4 SIMDs x waves_per_simd waves; block = 256 threads = 4 waves (one per SIMD)
  int blocks = 256 * waves_per_simd;
  float* out; CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, iters / 8, 1u);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, 3u);
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  double instr_per_simd = (double)waves_per_simd * iters * 16;
  double ns = ms * 1e6 / instr_per_simd;
  printf("%-28s waves/SIMD=%d  %.3f ms  %.3f ns per wave-instr per SIMD  (%.2f cyc @2.4GHz)\n", name, waves_per_simd, ms, ns, ns * 2.4);
  CHECK(hipFree(out));
  return ns;
}

#define RUN(OP, NAME) for (int w : {2, 4, 8}) run<OP>(NAME, w, 20000);

int main() {
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
  printf("%s CUs=%d clock=%d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
  RUN(0, "v_fma_f32") RUN(1, "v_pk_add_f16") RUN(2, "v_pk_mul_f16") RUN(3, "v_pk_fma_f16")
  RUN(4, "v_fma_mix_f32") RUN(5, "v_perm_b32") RUN(6, "v_and_b32") RUN(7, "v_dot2_f32_f16")
  RUN(8, "v_cvt_f32_f16") RUN(9, "v_add_f16") RUN(10, "mfma_f32_4x4x4f16") RUN(11, "mfma_f32_16x16x32_f16")
  RUN(13, "v_pk_fma_f32") RUN(12, "16 pk_add + 8 mfma4x4x4") RUN(20, "ds_read_b128 partitioned") RUN(21, "ds_read_b128 random")
  RUN(22, "ds_write_b128 rotating")
  return 0;
}
