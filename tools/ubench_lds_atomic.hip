// LDS float-atomic throughput on gfx950 (MI355X): could a bucket-accumulate formulation of the one-token GEMV
// (h[row][k] += f(s x) per index, then one 512 x 8 product per vector-row) beat 2 ds_read_b128 + 4 MFMAs per index?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_lds_atomic.hip -o tools/_build/ubench_lds_atomic
// Per wave-instruction (64 lanes): CU cycles of ds_add_f32 with random buckets (4 histograms of 256 per wave, as the
// kernel would have), conflict-free addresses, and ds_read_b128 gathers for comparison.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
enum { RANDOM = 0, LINEAR = 1, READ128 = 2, RANDOM_RTN = 3, PKF16 = 4, U32 = 5, U32LIN = 6, WRITE32 = 7 };

template <int MODE, int THREADS>
__global__ __launch_bounds__(THREADS) void k(float* out, unsigned long long* cyc, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 32768; i += THREADS) ((float*)smem)[i] = 0.f;
  __syncthreads();
  uint32_t r = (tid * 2654435761u) ^ (blockIdx.x * 40503u);
  const uint32_t base = (uint32_t)wave * (131072u / (THREADS / 64)) + (uint32_t)(lane & 3) * 1024u;   // 4 histograms per wave
  float v = 1.0f + lane;
  float4 acc = {0.f, 0.f, 0.f, 0.f};
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      r = r * 1664525u + 1013904223u;
      if constexpr (MODE == RANDOM) {
        const uint32_t a = base + ((r >> 10) & 255u) * 4u;
        asm volatile("ds_add_f32 %0, %1" :: "v"(a), "v"(v) : "memory");
      } else if constexpr (MODE == U32) {
        const uint32_t a = base + ((r >> 10) & 255u) * 4u;
        asm volatile("ds_add_u32 %0, %1" :: "v"(a), "v"(r) : "memory");
      } else if constexpr (MODE == U32LIN) {
        const uint32_t a = (uint32_t)wave * (131072u / (THREADS / 64)) + ((r >> 10) & 15u) * 256u + (uint32_t)lane * 4u;
        asm volatile("ds_add_u32 %0, %1" :: "v"(a), "v"(r) : "memory");
      } else if constexpr (MODE == WRITE32) {
        const uint32_t a = base + ((r >> 10) & 255u) * 4u;
        asm volatile("ds_write_b32 %0, %1" :: "v"(a), "v"(r) : "memory");
      } else if constexpr (MODE == PKF16) {
        const uint32_t a = base + ((r >> 10) & 255u) * 4u;
        asm volatile("ds_pk_add_f16 %0, %1" :: "v"(a), "v"(v) : "memory");
      } else if constexpr (MODE == RANDOM_RTN) {
        const uint32_t a = base + ((r >> 10) & 255u) * 4u;
        float o;
        asm volatile("ds_add_rtn_f32 %0, %1, %2" : "=v"(o) : "v"(a), "v"(v) : "memory");
        acc.x += o;
      } else if constexpr (MODE == LINEAR) {
        const uint32_t a = (uint32_t)wave * (131072u / (THREADS / 64)) + ((r >> 10) & 15u) * 256u + (uint32_t)lane * 4u;
        asm volatile("ds_add_f32 %0, %1" :: "v"(a), "v"(v) : "memory");
      } else {
        const uint32_t a = (((r >> 10) & 255u) << 8) | ((uint32_t)(lane & 15) << 4);
        typedef float f4_t __attribute__((ext_vector_type(4)));
        typedef __attribute__((address_space(3))) f4_t lds_f4;
        const f4_t q = *(const lds_f4*)(uintptr_t)a;
        acc.x += q[0]; acc.y += q[1]; acc.z += q[2]; acc.w += q[3];
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
  if (lane == 0) cyc[blockIdx.x * (THREADS / 64) + wave] = t1 - t0;
  out[blockIdx.x * THREADS + tid] = acc.x + acc.y + acc.z + acc.w + ((float*)smem)[tid];
}

template <int MODE, int THREADS>
static void run(const char* name, float* out, unsigned long long* cyc) {
  const int iters = 200, wgs = 256;
  auto kern = k<MODE, THREADS>;
  CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  hipLaunchKernelGGL(kern, dim3(wgs), dim3(THREADS), 131072, 0, out, cyc, iters);
  CHECK(hipDeviceSynchronize());
  hipLaunchKernelGGL(kern, dim3(wgs), dim3(THREADS), 131072, 0, out, cyc, iters);
  CHECK(hipDeviceSynchronize());
  const int nw = wgs * THREADS / 64;
  unsigned long long* h = (unsigned long long*)malloc(nw * 8);
  CHECK(hipMemcpy(h, cyc, nw * 8, hipMemcpyDeviceToHost));
  double sum = 0; for (int i = 0; i < nw; ++i) sum += (double)h[i];
  const double per_wave = sum / nw;
  const double waves = THREADS / 64;
  printf("%-72s %6.2f CU cycles per wave-instruction (%d waves per CU)\n", name, per_wave / (iters * 16.0 * waves), THREADS / 64);
  free(h);
}

int main() {
  float* out; unsigned long long* cyc;
  CHECK(hipMalloc(&out, 256 * 1024 * 4)); CHECK(hipMalloc(&cyc, 8192 * 8));
  run<RANDOM, 1024>("ds_add_f32, random bucket of 256, 4 histograms per wave", out, cyc);
  run<RANDOM, 512>("ds_add_f32, random bucket of 256, 4 histograms per wave", out, cyc);
  run<LINEAR, 1024>("ds_add_f32, conflict-free (lane = bank)", out, cyc);
  run<RANDOM_RTN, 1024>("ds_add_rtn_f32, random bucket", out, cyc);
  run<PKF16, 1024>("ds_pk_add_f16, random bucket", out, cyc);
  run<U32, 1024>("ds_add_u32, random bucket of 256, 4 histograms per wave", out, cyc);
  run<U32LIN, 1024>("ds_add_u32, conflict-free", out, cyc);
  run<WRITE32, 1024>("ds_write_b32, random bucket", out, cyc);
  run<READ128, 1024>("ds_read_b128 gather (16 lanes = 16 units of a random row: conflict-free)", out, cyc);
  return 0;
}
