// Steady-state HBM streaming rate of ONE long persistent launch (no kernel boundaries inside the
// measurement), by access pattern and bytes in flight (GPU box only):
//   hipcc --offload-arch=gfx950 -O3 -o tools/_build/ubench_stream2 tools/ubench_stream2.hip && tools/_build/ubench_stream2
// The buffer is 1 GiB (4x the Infinity Cache), read once per launch with 16-byte loads, as
// "layers" of 1024 rows x 16 KiB (the 8192^2 index tensor).
//   rows4  : workgroup = 4 rows of a layer; wave w, lane (chunk = l >> 2, row = l & 3): 4 x 256 B per wave-load,
//            16 KiB per workgroup-sweep as 4 x 4 KiB (the pattern of gemv_k256m / gemv_k256t)
//   rows1  : workgroup = 4 rows, one after the other; wave w reads 1 KiB contiguous: 16 KiB contiguous per sweep
//   linear : the whole buffer as one array, workgroup b reads 64 KiB pieces b, b + W, ...; 16 KiB contiguous per sweep
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int PAT, int DEPTH, bool NT, int THREADS>
__global__ __launch_bounds__(THREADS) void stream(const char* __restrict__ base, long long pieces, uint32_t* out) {
  // piece = 64 KiB = one row group (4 rows x 16 KiB) in layer order; THREADS x 16 B per sweep
  const int tid = threadIdx.x;
  constexpr int kSweepBytes = THREADS * 16;
  constexpr int kSweeps = 65536 / kSweepBytes;
  u32x4 acc = {0, 0, 0, 0};
  u32x4 q[DEPTH];
  auto addr = [&](long long piece, int s) -> const u32x4* {
    const char* p = base + piece * 65536;
    if (PAT == 0) {   // 4 rows x (THREADS / 4) chunks
      const int j = tid & 3, chunk = tid >> 2;
      return (const u32x4*)(p + (size_t)j * 16384 + (size_t)s * (kSweepBytes / 4) + chunk * 16);
    } else {          // contiguous
      return (const u32x4*)(p + (size_t)s * kSweepBytes + tid * 16);
    }
  };
  long long total = 0;
  for (long long pc = blockIdx.x; pc < pieces; pc += gridDim.x) total += kSweeps;
  // flat stream of (piece, sweep)
  long long ip = blockIdx.x; int is = 0;
  auto next = [&](long long& p, int& s) { if (++s == kSweeps) { s = 0; p += gridDim.x; } };
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) {
    const long long pp = ip < pieces ? ip : blockIdx.x;
    q[d] = NT ? __builtin_nontemporal_load(addr(pp, is)) : *addr(pp, is);
    next(ip, is);
  }
  for (long long k = 0; k < total; k += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      acc ^= q[d];
      const long long pp = ip < pieces ? ip : blockIdx.x;
      q[d] = NT ? __builtin_nontemporal_load(addr(pp, is)) : *addr(pp, is);
      next(ip, is);
    }
  }
  const uint32_t r = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
  if (r == 0x12345678u && pieces < 0) out[tid] = r;
}

int main() {
  const size_t bytes = 1ull << 30;
  char* buf; CK(hipMalloc(&buf, bytes)); CK(hipMemset(buf, 1, bytes));
  uint32_t* out; CK(hipMalloc(&out, 4096));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const long long pieces = bytes / 65536;
#define RUN(PAT, D, NT, T, WPC, NAME) { \
    for (int it = 0; it < 2; ++it) hipLaunchKernelGGL((stream<PAT, D, NT, T>), dim3(cus * WPC), dim3(T), 0, st, buf, pieces, out); \
    CK(hipStreamSynchronize(st)); CK(hipEventRecord(e0, st)); \
    for (int it = 0; it < 5; ++it) hipLaunchKernelGGL((stream<PAT, D, NT, T>), dim3(cus * WPC), dim3(T), 0, st, buf, pieces, out); \
    CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); \
    printf("%-58s %7.1f us per GiB  %6.0f GB/s  (%.3f of 8 TB/s; %.2f us per 16 MiB)\n", NAME, ms * 1e3 / 5, bytes / (ms / 5) / 1e6, \
           bytes / (ms / 5) / 1e6 / 8000, ms * 1e3 / 5 / 64); }
  RUN(0, 2, true, 1024, 1, "rows4  1024 thr, 2 in flight (32 KiB / CU), nt")
  RUN(0, 4, true, 1024, 1, "rows4  1024 thr, 4 in flight (64 KiB / CU), nt")
  RUN(0, 8, true, 1024, 1, "rows4  1024 thr, 8 in flight (128 KiB / CU), nt")
  RUN(0, 4, false, 1024, 1, "rows4  1024 thr, 4 in flight, plain loads")
  RUN(1, 2, true, 1024, 1, "contig 1024 thr, 2 in flight, nt")
  RUN(1, 4, true, 1024, 1, "contig 1024 thr, 4 in flight, nt")
  RUN(1, 8, true, 1024, 1, "contig 1024 thr, 8 in flight, nt")
  RUN(1, 4, false, 1024, 1, "contig 1024 thr, 4 in flight, plain loads")
  RUN(0, 4, true, 512, 2, "rows4  2 x 512 thr per CU, 4 in flight, nt")
  RUN(1, 4, true, 512, 2, "contig 2 x 512 thr per CU, 4 in flight, nt")
  RUN(1, 8, true, 256, 4, "contig 4 x 256 thr per CU, 8 in flight, nt")
  RUN(1, 8, true, 256, 8, "contig 8 x 256 thr per CU, 8 in flight, nt")
  RUN(0, 8, true, 256, 8, "rows4  8 x 256 thr per CU, 8 in flight, nt")
  return 0;
}
