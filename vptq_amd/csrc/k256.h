// Launch parameters shared by the two k=256 GEMV kernels (gemv_k256.hip: packed-f16 VALU
// accumulate; gemv_k256m.hip: MFMA accumulate).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <hip/hip_runtime.h>

namespace vptq {

constexpr int kMaxGroup = 32;
constexpr int kTableBytes = 65536;        // gemv_k256: 256 rows x 256 B
constexpr int kScratchOff = kTableBytes;

struct K256Layer {
  const uint32_t* idx;    // [N, row_words]
  const uint32_t* cent;   // [256, 8] as 4 dwords per entry
  const uint32_t* rcent;  // [256, 8]
  const uint16_t* x;      // [tokens, I]
  uint16_t* y;            // [tokens, O]
  const uint16_t* scale;  // [I]
  const uint16_t* wbias;  // [I]
  const uint16_t* bias;   // [O] or null
  const uint16_t* perm;   // [I] or null
  const char* pf;         // read-ahead range (next layer's indices) or null
  long long pf_bytes;
  int N, G, O, row_words;
  int wgs;       // gemv_k256m: persistent workgroups walking this layer's row groups
  int pf_chunk;  // read-ahead stride per workgroup in bytes (multiple of 128)
  int pf_len;    // bytes actually touched per workgroup (<= pf_chunk)
  int slots;      // gemv_k256m: cross-wave partial-sum slots in LDS (bits 0-7) | bit 8: VPTQ_GEMV_SELECTIVE
};

// K256Params::tokens: token count in the low 16 bits; kOutF32Bit set = y is float32
// [tokens, O] (VPTQ_GEMV_OUT_F32: un-rounded sums, e.g. the partial outputs of a row-parallel
// shard that are rounded once after the all-reduce)
constexpr int kOutF32Bit = 1 << 16;

struct K256Params {
  int n_layers;
  int tokens;
  K256Layer layer[kMaxGroup];
};

// This workgroup's layer (blockIdx.y) and the token count, fetched from the kernel-argument
// segment with ONE batch of scalar loads and one wait (inside the same asm statement, so no
// register is read before it has landed).  Left to the compiler, the fields are loaded where
// they are first used: four dependent kernarg round trips before the first vector load can be
// issued (measured: 1.3-1.7 us median, tools/trace_k256m.py).
static __device__ __forceinline__ K256Layer load_layer_args(int& tokens) {
  static_assert(sizeof(K256Layer) == 120 && offsetof(K256Params, layer) == 8, "kernarg layout");
  typedef int i16_t __attribute__((ext_vector_type(16)));
  typedef int i8_t __attribute__((ext_vector_type(8)));
  typedef int i4_t __attribute__((ext_vector_type(4)));
  typedef int i2_t __attribute__((ext_vector_type(2)));
  const char __attribute__((address_space(4)))* base =
      (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
  const char __attribute__((address_space(4)))* lp = base + 8 + (size_t)blockIdx.y * sizeof(K256Layer);
  i16_t a; i8_t b; i4_t c; i2_t d; int tk;
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile(
      "s_load_dwordx16 %0, %5, 0x0\n\t"
      "s_load_dwordx8 %1, %5, 0x40\n\t"
      "s_load_dwordx4 %2, %5, 0x60\n\t"
      "s_load_dwordx2 %3, %5, 0x70\n\t"
      "s_load_dword %4, %6, 0x4\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&s"(a), "=&s"(b), "=&s"(c), "=&s"(d), "=&s"(tk)
      : "s"(lp), "s"(base)
      : "memory");
#else
  a = i16_t{}; b = i8_t{}; c = i4_t{}; d = i2_t{}; tk = 0; (void)lp;
#endif
  K256Layer L;
  __builtin_memcpy((char*)&L, &a, 64);
  __builtin_memcpy((char*)&L + 64, &b, 32);
  __builtin_memcpy((char*)&L + 96, &c, 16);
  __builtin_memcpy((char*)&L + 112, &d, 8);
  L.idx = as_global(L.idx); L.cent = as_global(L.cent); L.rcent = as_global(L.rcent);
  L.x = as_global(L.x); L.y = as_global(L.y); L.scale = as_global(L.scale);
  L.wbias = as_global(L.wbias); L.bias = as_global(L.bias); L.perm = as_global(L.perm);
  L.pf = as_global(L.pf);
  tokens = tk;
  return L;
}

// -DVPTQ_K256_TRACE (make trace; tools/trace_k256m.py): every wave stamps the 100 MHz wall
// clock at its phase boundaries into the buffer passed as the read-ahead range:
// [workgroup][wave][8] x u64.
#ifdef VPTQ_K256_TRACE
#define K256_STAMP(nwaves, k, dep)                                                              \
  do {                                                                                          \
    unsigned long long t_;                                                                      \
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) : "v"(dep) : "memory");  \
    if (lane == 0) ((unsigned long long*)Ly.pf)[((size_t)bid * (nwaves) + wave) * 8 + (k)] = t_; \
  } while (0)
#else
#define K256_STAMP(nwaves, k, dep) do { } while (0)
#endif

// gemv_k256m.hip
constexpr int kMRows = 4;  // vector-rows per workgroup of the MFMA kernel
bool gemv_k256m_supported(int tok, bool f16, bool fast, int max_cols, bool perm);
int gemv_k256m_row_groups(int n_rows);
hipError_t launch_gemv_k256m(K256Params& P, int tok, bool f16, bool fast, int max_cols, bool perm,
                             hipStream_t st, bool selective = false);
bool gemv_k256m_selective_ok(const int* n_rows, int n, bool f16, int tok, int max_cols, bool perm);
// most row groups any workgroup of one launch of these layers walks (the launch's duration in units of one row group)
int gemv_k256m_launch_units(const int* n_rows, int n);

}  // namespace vptq
