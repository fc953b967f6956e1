// Prints the operand layout of v_mfma_f32_4x4x4_16b_f16 on the device it runs on.
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
// First operand: lane l holds {4l, 4l+1, 4l+2, 4l+3}; second operand: lane l holds e_(l&3) * (l&3 + 1).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out, int swap) {
  const int l = threadIdx.x;
  h4 a, b;
  for (int q = 0; q < 4; ++q) { a[q] = (_Float16)(float)(4 * l + q); b[q] = (_Float16)((q == (l & 3)) ? (float)((l & 3) + 1) : 0.f); }
  f4 d = {0, 0, 0, 0};
  d = swap ? __builtin_amdgcn_mfma_f32_4x4x4f16(b, a, d, 0, 0, 0) : __builtin_amdgcn_mfma_f32_4x4x4f16(a, b, d, 0, 0, 0);
  for (int q = 0; q < 4; ++q) out[l * 4 + q] = d[q];
}
int main() {
  float* d; hipMalloc(&d, 64 * 4 * 4);
  float h[256];
  for (int swap = 0; swap < 2; ++swap) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, swap);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("swap=%d (weights %s operand)\n", swap, swap ? "second" : "first");
    for (int l = 0; l < 12; ++l) printf("  lane %2d: %6.0f %6.0f %6.0f %6.0f\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
    // hypothesis H1: lane j (of block b), reg i = A_(4b+i)[j] * (j+1)
    int ok1 = 1, ok2 = 1;
    for (int l = 0; l < 64; ++l) for (int i = 0; i < 4; ++i) {
      int b = l >> 2, j = l & 3;
      if (h[l*4+i] != (float)((4 * (4*b+i) + j) * (j + 1))) ok1 = 0;   // row i's weight t=j, scaled by lane j's x
      if (h[l*4+i] != (float)((4 * (4*b+j) + i) * (i + 1))) ok2 = 0;   // lane j's own weights, t = i
    }
    printf("  H1 (lane j reg i = weights of lane 4b+i, element j): %s\n  H2 (lane j reg i = own weights element i, scaled by lane i's x): %s\n", ok1 ? "YES" : "no", ok2 ? "YES" : "no");
  }
  return 0;
}
