// Feasibility of a bucket-accumulate one-token GEMV on gfx950 (MI355X): the chain kernel (gemv_k256c.hip) runs at the
// 1400 W package power limit (shader clock 1.62 of 2.4 GHz); would "h[row][k] += fixed-point(s x) per index with
// ds_add_u32, then one 512 x 8 product per vector-row" need less energy per index than 2 ds_read_b128 + 4 MFMAs?
// Skeleton of the histogram phase only: the 8192^2 index stream of the real kernel (persistent 256 x 1024 threads,
// lane = (8-column chunk, vector-row), two 16-byte loads per lane and sweep, 3 sweeps in flight), per index two
// ds_add_u32 into the wave's own histograms (2 x 4 vector-rows x 2 codebooks x 256 buckets x 4 B = 16 KiB per wave...
// here 8 KiB: one set for both row subgroups - the LDS holds 128 KiB), histogram cleared per row group.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_bucket.hip -o tools/_build/ubench_bucket && tools/_build/ubench_bucket [seconds]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>   // 0: adds, 1: stream only, 2: adds of the main index only
__global__ __launch_bounds__(1024) void bucket(const char* __restrict__ base, int layers, uint32_t* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int D = 3;
  // a layer = 1024 vector-rows x 8192 columns x 2 B; a row group = 8 vector-rows (2 subgroups of 4), 4 sweeps of 2048
  // columns; the workgroup takes row groups b, b + W, ... of every layer
  const int jrow = lane & 3, chunk = lane >> 2;
  const uint32_t hist = (uint32_t)wave * 8192u + (uint32_t)jrow * 2048u;   // [codebook 2][256] per vector-row
  u32x4 q[D][2];
  auto addr = [&](long long sweep, int sub) -> const u32x4* {   // sweep index in the workgroup's flat stream
    const long long rgq = sweep >> 2; const int s = (int)(sweep & 3);
    const long long rg = (long long)blockIdx.x + rgq * gridDim.x;   // global row group (over all layers)
    const char* p = base + rg * 131072 + (size_t)(sub * 4 + jrow) * 16384 + (size_t)s * 4096 + (size_t)wave * 256 + chunk * 16;
    return (const u32x4*)p;
  };
  const long long n_rg_total = (long long)layers * 128;
  long long my = 0;
  for (long long rg = blockIdx.x; rg < n_rg_total; rg += gridDim.x) ++my;
  const long long total = my * 4;
  for (int i = tid; i < 32768; i += 1024) ((uint32_t*)smem)[i] = 0u;
  __syncthreads();
  long long is = 0;
#pragma unroll
  for (int d = 0; d < D; ++d) {
    const long long s = is < total ? is : total - 1;
    q[d][0] = __builtin_nontemporal_load(addr(s, 0)); q[d][1] = __builtin_nontemporal_load(addr(s, 1));
    ++is;
  }
  uint32_t val[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) val[i] = 1000u + lane * 8 + i;
  uint32_t accx = 0;
  for (long long k = 0; k < total; k += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      if (MODE == 1) { accx ^= q[d][0][0] ^ q[d][1][3]; }
      else {
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const uint32_t w = q[d][sub][u >> 1];
            const uint32_t sh = (u & 1) * 16;
            const uint32_t a0 = hist + (((w >> sh) & 255u) << 2);
            const uint32_t a1 = hist + 1024u + (((w >> (sh + 8)) & 255u) << 2);
            asm volatile("ds_add_u32 %0, %1" :: "v"(a0), "v"(val[u]) : "memory");
            if (MODE == 0) asm volatile("ds_add_u32 %0, %1" :: "v"(a1), "v"(val[u]) : "memory");
          }
        }
      }
      const long long s = is < total ? is : total - 1;
      q[d][0] = __builtin_nontemporal_load(addr(s, 0)); q[d][1] = __builtin_nontemporal_load(addr(s, 1));
      ++is;
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if ((accx == 0x12345678u && layers < 0) || layers == -5) out[tid] = accx + ((uint32_t*)smem)[tid];
}

int main(int argc, char** argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 0.0;
  const int mode_loop = argc > 2 ? atoi(argv[2]) : 0;
  const int layers = 32;
  const size_t bytes = (size_t)layers * 16 * 1024 * 1024;
  char* buf; CK(hipMalloc(&buf, bytes));
  {   // random index bytes
    uint32_t* h = (uint32_t*)malloc(bytes);
    uint32_t r = 12345u;
    for (size_t i = 0; i < bytes / 4; ++i) { r = r * 1664525u + 1013904223u; h[i] = r ^ (r >> 13); }
    CK(hipMemcpy(buf, h, bytes, hipMemcpyHostToDevice)); free(h);
  }
  uint32_t* out; CK(hipMalloc(&out, 4096));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipFuncSetAttribute((const void*)bucket<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  CK(hipFuncSetAttribute((const void*)bucket<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  CK(hipFuncSetAttribute((const void*)bucket<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
#define RUN(M, NAME) { \
    for (int it = 0; it < 2; ++it) hipLaunchKernelGGL((bucket<M>), dim3(256), dim3(1024), 131072, st, buf, layers, out); \
    CK(hipStreamSynchronize(st)); CK(hipEventRecord(e0, st)); \
    for (int it = 0; it < 5; ++it) hipLaunchKernelGGL((bucket<M>), dim3(256), dim3(1024), 131072, st, buf, layers, out); \
    CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); \
    printf("%-64s %7.2f us per 8192^2 layer (%5.0f GB/s of index words)\n", NAME, ms * 1e3 / 5 / layers, bytes / (ms / 5) / 1e6); }
  if (seconds <= 0.0) {
    RUN(1, "stream only")
    RUN(2, "stream + 1 ds_add_u32 per index (main codebook only)")
    RUN(0, "stream + 2 ds_add_u32 per index")
  } else {   // back to back for `seconds` (power probe)
    auto t0 = std::chrono::steady_clock::now();
    long long n = 0;
    CK(hipEventRecord(e0, st));
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
      for (int it = 0; it < 20; ++it) {
        if (mode_loop == 1) hipLaunchKernelGGL((bucket<1>), dim3(256), dim3(1024), 131072, st, buf, layers, out);
        else hipLaunchKernelGGL((bucket<0>), dim3(256), dim3(1024), 131072, st, buf, layers, out);
      }
      n += 20;
      CK(hipStreamSynchronize(st));
    }
    CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st)); float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("mode %d back to back: %.2f us per layer\n", mode_loop, ms * 1e3 / n / layers);
  }
  return 0;
}
