// LDS gather / scatter-add throughput on gfx950 (MI355X) for the address patterns the GEMV
// kernels could use.  Build & run on the GPU box:
//   hipcc --offload-arch=gfx950 -O2 tools/ubench_lds.hip -o /tmp/ubench_lds && /tmp/ubench_lds
// Every wave issues 16 DS operations per iteration on addresses derived from a per-lane
// pseudo-random entry number k in [0,256) (different every operation), laid out as:
//   gather (ds_read_b128, 16-byte entries):
//     g16   16 replicas, lane reads replica lane&15           (conflict free, 64 KiB)
//     g8p    8 replicas in 8 fixed slots, lane&7              (the round-1 kernel, 2-way, 64 KiB incl. residual)
//     g8r    8 replicas, slot = 2*(lane&7) + (k&1)            (balls-in-bins, 32 KiB)
//     g4r    4 replicas, slot = 4*(lane&3) + (k&3)            (16 KiB)
//     g1     no replicas                                      (4 KiB)
//     g8x    two tables x 8 replicas in one 16-slot row: lanes with bit 3 clear read table A
//            (slots 0-7), the others table B (slots 8-15)       (conflict free, 64 KiB for BOTH)
//   scatter-add (ds_add_f32, 4-byte buckets):
//     a32   32 replicas, lane&31 owns a bank                  (conflict free, 32 KiB)
//     a16   16 replicas, bank = (lane&15) + 16*(k&1)          (16 KiB)
//     a8     8 replicas, bank = (lane&7) + 8*(k&3)            (8 KiB)
//     a4     4 replicas                                       (4 KiB)
//     a1     no replicas                                      (1 KiB, same-address collisions)
// Reported: LDS cycles per wave-instruction per CU at the given clock (all 4 SIMDs issuing).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

typedef unsigned u4 __attribute__((ext_vector_type(4)));

template <int OP>
__device__ __forceinline__ unsigned addr_of(unsigned k, unsigned lane) {
  switch (OP) {
    case 0: return (k << 8) | ((lane & 15) << 4);                                   // g16
    case 1: return (k << 8) | ((lane & 7) << 4);                                    // g8p
    case 2: return ((k >> 1) << 8) | ((lane & 7) << 5) | ((k & 1) << 4);            // g8r
    case 3: return ((k >> 2) << 8) | ((lane & 3) << 6) | ((k & 3) << 4);            // g4r
    case 4: return k << 4;                                                          // g1
    case 5: return (k << 8) | ((((lane >> 3) & 1) * 8 + (lane & 7)) << 4);          // g8x: 16 slots from 2 x 8 replicas
    case 10: return (k << 7) | ((lane & 31) << 2);                                  // a32
    case 11: return (k << 6) | ((lane & 15) << 2);                                  // a16: dword = 16k + rho -> bank = rho + 16*(k&1)
    case 12: return (k << 5) | ((lane & 7) << 2);                                   // a8
    case 13: return (k << 4) | ((lane & 3) << 2);                                   // a4
    case 14: return k << 2;                                                         // a1
    case 20: return (k << 7) | ((lane & 31) << 2);                                  // pk_add_f16 a32
    case 21: return (k << 5) | ((lane & 7) << 2);                                   // pk_add_f16 a8
    case 30: return (k << 7) | ((lane & 31) << 2);                                  // add_rtn a32
  }
  return 0;
}

template <int OP>
__global__ __launch_bounds__(512) void k(float* out, int iters, unsigned seed) {
  extern __shared__ __attribute__((aligned(16))) unsigned lds[];
  for (int i = threadIdx.x; i < 16384; i += 512) lds[i] = 0;
  __syncthreads();
  const unsigned t = threadIdx.x + blockIdx.x * 512;
  const unsigned lane = threadIdx.x & 63;
  unsigned r = (t * 2654435761u + seed) ^ ((t * 40503u) >> 3);
  r ^= r >> 13; r *= 0x5bd1e995u; r ^= r >> 15;
  unsigned acc = 0;
  float facc = 0.f;
  const float val = 1.0f + (float)(t & 3);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      r += 0x9e3779b9u;                       // per-lane start is random; common stride
      const unsigned kk = (r >> 11) & 255u;
      const unsigned ad = addr_of<OP>(kk, lane);
      if (OP < 10) {
        u4 v;
        asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(ad) : "memory");
        if (j == 15) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); acc ^= v.x; }
      } else if (OP < 20) {
        asm volatile("ds_add_f32 %0, %1" : : "v"(ad), "v"(val) : "memory");
      } else if (OP < 30) {
        asm volatile("ds_pk_add_f16 %0, %1" : : "v"(ad), "v"(0x3c003c00u) : "memory");
      } else {
        float o;
        asm volatile("ds_add_rtn_f32 %0, %1, %2" : "=v"(o) : "v"(ad), "v"(val) : "memory");
        if (j == 15) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); facc += o; }
      }
    }
    if (OP >= 10 && OP < 30) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  __syncthreads();
  if (acc == 0x12345678u || facc == 1234.5f || lds[threadIdx.x] == 0xdeadbeefu) out[t] = 1.f;
}

template <int OP>
void run(const char* name, int waves_per_simd, int iters, double ghz) {
  int blocks = 256 * waves_per_simd;
  float* out; CHECK(hipMalloc(&out, (size_t)blocks * 512 * 4));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  CHECK(hipFuncSetAttribute((const void*)k<OP>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  // LDS per block chosen so that exactly waves_per_simd blocks fit a CU
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(512), 65536, 0, out, iters / 8, 1u);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(512), 65536, 0, out, iters, 3u);
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  // per CU: blocks x 8 waves x iters x 16 ops
  double ops_per_cu = 8.0 * waves_per_simd * iters * 16;  // waves_per_simd = blocks per CU here (8 waves each)
  double ns = ms * 1e6 / ops_per_cu;
  printf("%-22s blocks/CU=%d  %8.3f ms  %6.3f ns per wave-instr per CU  (%5.2f cyc @%.1fGHz)\n", name,
         waves_per_simd, ms, ns, ns * ghz, ghz);
  CHECK(hipFree(out));
}

#define RUN(OP, NAME) for (int w : {1, 2}) run<OP>(NAME, w, 4000, 2.4);

int main() {
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
  printf("%s CUs=%d clock=%d kHz (64 KiB LDS per 512-thread block: 1-2 blocks = 2-4 waves per SIMD)\n", p.gcnArchName,
         p.multiProcessorCount, p.clockRate);
  RUN(0, "gather b128 g16") RUN(1, "gather b128 g8p") RUN(2, "gather b128 g8r") RUN(3, "gather b128 g4r")
  RUN(4, "gather b128 g1") RUN(5, "gather b128 g8x")
  RUN(10, "ds_add_f32 a32") RUN(11, "ds_add_f32 a16") RUN(12, "ds_add_f32 a8") RUN(13, "ds_add_f32 a4")
  RUN(14, "ds_add_f32 a1") RUN(20, "ds_pk_add_f16 a32") RUN(21, "ds_pk_add_f16 a8") RUN(30, "ds_add_rtn_f32 a32")
  return 0;
}
