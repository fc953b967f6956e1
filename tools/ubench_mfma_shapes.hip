// Issue cost of the MFMA shapes a one-token dequant-GEMV could accumulate with ("why no larger
// shape"): s_memtime cycles per instruction on ONE SIMD, 1 and 4 waves, four independent accumulators each.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma_shapes.hip -o /tmp/ubench_mfma_shapes && /tmp/ubench_mfma_shapes
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16 __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));

static __device__ __forceinline__ unsigned long long now() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}

template <int SHAPE>
__global__ void k(const h16* src, f4* dst, unsigned long long* cyc, int n) {
  const h16 bv = src[threadIdx.x & 63];
  const h8 a8 = {bv[0], bv[1], bv[2], bv[3], bv[4], bv[5], bv[6], bv[7]};
  const h4 a4 = {bv[0], bv[1], bv[2], bv[3]}, b4 = {bv[4], bv[5], bv[6], bv[7]};
  const h8 b8 = {bv[8], bv[9], bv[10], bv[11], bv[12], bv[13], bv[14], bv[15]};
  const int ix = threadIdx.x & 0x4444;
  f4 acc[4] = {};
  const unsigned long long t0 = now();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if constexpr (SHAPE == 0) acc[u] = __builtin_amdgcn_mfma_f32_4x4x4f16(a4, b4, acc[u], 0, 0, 0);
      if constexpr (SHAPE == 1) acc[u] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc[u], 0, 0, 0);
      if constexpr (SHAPE == 2) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc[u], 0, 0, 0);
      if constexpr (SHAPE == 3) acc[u] = __builtin_amdgcn_smfmac_f32_16x16x64_f16(a8, bv, acc[u], ix, 0, 0);
    }
  }
  const unsigned long long t1 = now();
  dst[threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
  if ((threadIdx.x & 63) == 0) { cyc[2 * (threadIdx.x >> 6)] = t0; cyc[2 * (threadIdx.x >> 6) + 1] = t1; }
}

template <int SHAPE>
static void run(const char* name, int halves, const h16* src, f4* dst, unsigned long long* cyc) {
  const int n = 4096;
  for (int waves : {1, 4}) {   // 4 waves of one workgroup of 256 threads = one per SIMD; 1024 threads = 4 per SIMD
    const int threads = waves == 1 ? 64 : 1024;
    hipLaunchKernelGGL(k<SHAPE>, dim3(1), dim3(threads), 0, 0, src, dst, cyc, n);
    hipLaunchKernelGGL(k<SHAPE>, dim3(1), dim3(threads), 0, 0, src, dst, cyc, n);
    hipDeviceSynchronize();
    unsigned long long h[32];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    // the SIMDs serve their waves oldest-first: one wave's own elapsed time says nothing about the pipe -
    // first start to last end over all waves of the workgroup (16 waves = 4 per SIMD)
    const int nw = threads / 64;
    unsigned long long lo = ~0ull, hi = 0;
    for (int w = 0; w < nw; ++w) { lo = h[2 * w] < lo ? h[2 * w] : lo; hi = h[2 * w + 1] > hi ? h[2 * w + 1] : hi; }
    const double per = (double)(hi - lo) / (n * 4.0 * (waves == 1 ? 1 : 4));
    printf("%-34s %d wave(s) per SIMD: %8.3f s_memtime cycles per instruction and SIMD, %d gathered halves per lane\n",
           name, waves, per, halves);
  }
}

int main() {
  h16* src; f4* dst; unsigned long long* cyc;
  hipMalloc(&src, 64 * sizeof(h16)); hipMalloc(&dst, 1024 * sizeof(f4)); hipMalloc(&cyc, 32 * 8);
  hipMemset(src, 0, 64 * sizeof(h16));
  run<0>("v_mfma_f32_4x4x4_16b_f16", 4, src, dst, cyc);
  run<1>("v_mfma_f32_16x16x16_f16", 4, src, dst, cyc);
  run<2>("v_mfma_f32_16x16x32_f16", 8, src, dst, cyc);
  run<3>("v_smfmac_f32_16x16x64_f16 (2:4 A)", 16, src, dst, cyc);
  return 0;
}
