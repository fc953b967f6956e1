// Floor of ONE decode-GEMV launch that does nothing but stream its index words (GPU box only):
//   hipcc --offload-arch=gfx950 -O3 -o tools/_build/ubench_stream tools/ubench_stream.hip
//   tools/_build/ubench_stream [rows=1024] [row_bytes=16384]
// A ring of R distinct buffers (R x bytes >= 512 MiB: neither L2 nor the 256 MiB Infinity Cache
// holds them), one launch per buffer, the R launches captured in a hipGraph and replayed;
// us per launch = HIP-event time / launches, i.e. INCLUDING the kernel boundary - the same
// clock bench.py uses for gemv_k256m_kernel.  Every variant reads each byte once with 16-byte
// loads and folds it into one word per thread (stored only if a never-true condition holds),
// so what is measured is dispatch + HBM stream + drain, no LDS image, no gathers, no MFMA:
//   persistent : 256 workgroups x 1024 threads, workgroup b streams rows 4b..4b+3 in sweeps
//                of 16 KiB, DEPTH sweeps in flight (the access pattern of gemv_k256m_kernel)
//   flat       : rows*row_bytes/16/256 workgroups x 256 threads, one load per thread x 4
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int DEPTH, bool NT>
__global__ __launch_bounds__(1024) void stream_persistent(const char* __restrict__ base, int rows, int row_bytes,
                                                          uint32_t* out) {
  const int tid = threadIdx.x, j = tid & 3, chunk = tid >> 2;  // lane = (16-byte chunk, row of the group)
  const int sweeps = row_bytes / 4096;  // 256 chunks x 16 B per row and sweep
  u32x4 acc = {0, 0, 0, 0};
  for (int rg = blockIdx.x; rg * 4 < rows; rg += gridDim.x) {
    const char* p = base + (size_t)(rg * 4 + j) * row_bytes + chunk * 16;
    u32x4 q[DEPTH];
#pragma unroll
    for (int s = 0; s < DEPTH; ++s) {
      const int sc = s < sweeps ? s : sweeps - 1;   // fewer sweeps than slots: re-read the last one
      q[s] = NT ? __builtin_nontemporal_load((const u32x4*)(p + sc * 4096)) : *(const u32x4*)(p + sc * 4096);
    }
    for (int s = 0; s < sweeps; s += DEPTH) {
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        acc ^= q[d];
        const int nxt = s + d + DEPTH;
        if (nxt < sweeps)
          q[d] = NT ? __builtin_nontemporal_load((const u32x4*)(p + nxt * 4096)) : *(const u32x4*)(p + nxt * 4096);
      }
    }
  }
  const uint32_t r = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
  if (r == 0x12345678u && rows < 0) out[tid] = r;
}

template <bool NT>
__global__ __launch_bounds__(256) void stream_flat(const char* __restrict__ base, size_t bytes, uint32_t* out) {
  const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 16;
  const size_t quarter = bytes / 4;
  u32x4 acc = {0, 0, 0, 0};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const u32x4* p = (const u32x4*)(base + k * quarter + i);
    acc ^= NT ? __builtin_nontemporal_load(p) : *p;
  }
  const uint32_t r = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
  if (r == 0x12345678u && bytes == 1) out[threadIdx.x] = r;
}

__global__ void empty_kernel(uint32_t* out) { if (out == nullptr) __builtin_trap(); }

template <typename F>
static double time_ring(F launch, int R, int iters, hipStream_t st) {
  for (int i = 0; i < R; ++i) launch(i);
  CK(hipStreamSynchronize(st));
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
  for (int i = 0; i < R; ++i) launch(i);
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  CK(hipGraphLaunch(ge, st));
  CK(hipStreamSynchronize(st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, st));
  for (int it = 0; it < iters; ++it) CK(hipGraphLaunch(ge, st));
  CK(hipEventRecord(e1, st));
  CK(hipStreamSynchronize(st));
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  return ms * 1e3 / (iters * (double)R);
}

int main(int argc, char** argv) {
  const int rows = argc > 1 ? atoi(argv[1]) : 1024;
  const int row_bytes = argc > 2 ? atoi(argv[2]) : 16384;
  const size_t bytes = (size_t)rows * row_bytes;
  const int R = (int)((512ull << 20) / bytes) > 2 ? (int)((512ull << 20) / bytes) : 2;
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  printf("%s CUs=%d; %d rows x %d B = %.2f MiB per launch, ring of %d buffers\n", prop.gcnArchName,
         prop.multiProcessorCount, rows, row_bytes, bytes / 1048576.0, R);
  std::vector<char*> bufs(R);
  for (int i = 0; i < R; ++i) { CK(hipMalloc(&bufs[i], bytes)); CK(hipMemset(bufs[i], i + 1, bytes)); }
  uint32_t* out; CK(hipMalloc(&out, 4096));
  hipStream_t st; CK(hipStreamCreate(&st));
  const int cus = prop.multiProcessorCount, iters = 20;
  auto report = [&](const char* name, double us) {
    printf("%-44s %7.2f us per launch  %6.0f GB/s  (%.3f of 8 TB/s)\n", name, us, bytes / us / 1e3, bytes / us / 1e3 / 8000);
  };
  report("empty kernel (boundary only)", time_ring([&](int) { hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, st, out); }, R, iters, st));
  for (int rep = 0; rep < 2; ++rep) {
#define PERS(D, NT, NAME) report(NAME, time_ring([&](int i) { hipLaunchKernelGGL((stream_persistent<D, NT>), dim3(cus), dim3(1024), 0, st, bufs[i], rows, row_bytes, out); }, R, iters, st));
    PERS(1, false, "persistent 256 x 1024, 1 sweep in flight")
    PERS(2, false, "persistent 256 x 1024, 2 sweeps in flight")
    PERS(4, false, "persistent 256 x 1024, 4 sweeps in flight")
    PERS(2, true, "persistent 256 x 1024, 2 sweeps, nt loads")
    PERS(4, true, "persistent 256 x 1024, 4 sweeps, nt loads")
    report("flat, 256-thread workgroups, 4 loads / thread", time_ring([&](int i) { hipLaunchKernelGGL((stream_flat<false>), dim3((unsigned)(bytes / 64 / 256)), dim3(256), 0, st, bufs[i], bytes, out); }, R, iters, st));
    report("flat, nt loads", time_ring([&](int i) { hipLaunchKernelGGL((stream_flat<true>), dim3((unsigned)(bytes / 64 / 256)), dim3(256), 0, st, bufs[i], bytes, out); }, R, iters, st));
  }
  return 0;
}
