// What ds_read_b64_tr_b16 (gfx950) does, printed: LDS halfword i holds the value i; lane l reads
// the 8-byte chunk at byte address 8 l.  For every result lane: which source chunk (= source lane)
// and which element of it each of its 4 result elements came from.
//   hipcc --offload-arch=gfx950 -O2 tools/tr_probe.hip -o tools/_build/tr_probe && tools/_build/tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void k(s4* out) {
  __shared__ uint16_t lds[64 * 4];
  for (int i = threadIdx.x; i < 256; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  typedef __attribute__((address_space(3))) s4 lds_s4;
  out[threadIdx.x] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(lds + threadIdx.x * 4));
}
int main() {
  s4* d; hipMalloc(&d, 64 * sizeof(s4));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  s4 h[64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int var0 = 1, var1 = 1;
  for (int l = 0; l < 64; ++l) {
    printf("lane %2d:", l);
    for (int e = 0; e < 4; ++e) {
      const int v = (uint16_t)h[l][e], src = v / 4, el = v % 4;
      printf("  [%d] <- lane %2d elem %d", e, src, el);
      const int g = l & ~15, i = l & 15;
      if (!(src == g + 4 * e + i / 4 && el == i % 4)) var0 = 0;
      if (!(src == g + e + 4 * (i / 4) && el == i % 4)) var1 = 0;
    }
    printf("\n");
  }
  printf("convention: %s\n", var0 ? "0 (source lane 4e + c -> result lane 4c + m, element e)"
                           : var1 ? "1 (source lane e + 4c)" : "NEITHER");
  return 0;
}
