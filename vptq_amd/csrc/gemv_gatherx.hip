// Fused dequant + GEMV for every format the specialised kernels do not take: any vector length the
// reference dispatches (2, 4, 6, 8, 10, 12, 16: csrc/quant_gemv.cu:210-233), any main codebook up to
// 65536 entries, any residual codebook (none ... 65536 entries, i.e. ANY total index width
// T = index_bits + res_bits <= 32, not only 16 / 24 / 32), one or several codebook groups, outlier
// columns whose codebook has the layer's vector length or vector length 4 (the only one the
// reference's kernel takes, csrc/quant_gemv.cu:186-189) under a vector length of 8 / 12 / 16.  These
// are the formats of the larger published checkpoints ("v16-k65536-65536", "v16-k65536-32768",
// "v12-k65536-4096", "v8-k32768-0", "v6-k4096-0", ...), which ran on gemv_generic.hip (one dependent
// index -> gather chain per element, 0.04 of the roofline).
//
// Replaces WqA16WithOutliers_PackIndice for those template cases (reference
// csrc/kernels/quant_gemv.cuh:11-186; its dispatch over index_bits x res_bits x vector length:
// csrc/quant_gemv.cu:42-132).
//
// Same decomposition as gemv_gather.hip - a 256-thread workgroup owns one complete vector-row over
// all input columns, codebook rows are gathered from L2 with 16-byte loads, all gathers of a piece
// are in flight before the first is used, the reference's roundings, fp32 accumulation, lane-swap
// reduce-scatter - with an index path for arbitrary T:
//  * a lane takes 4 consecutive elements = 4 T bits, a window of <= 5 consecutive 32-bit words that
//    starts at bit 4 T lane (a multiple of 4 bits); neighbouring lanes read neighbouring windows.
//    The window is shifted to bit 0 once (v_alignbit_b32 with the lane's offset); element e then
//    sits at bit e T, a WAVE-UNIFORM position: the word selection is scalar control flow, not a
//    per-lane register index.  A window that would run past the row end is read word by word with
//    the word number clamped (the bits that matter are inside the row by construction).
//  * vector length 16: an entry is two 16-byte loads, 16 accumulators per token; vector length 12:
//    a 16-byte + an 8-byte load (24-byte entries are 8-byte aligned), 12 accumulators, padded to 16
//    for the wave reduction; 10: 16 + 4 bytes; 6: one 12-byte load; 4 / 2: one 8- / 4-byte load
//    (accumulators padded to 8).
#include "common.h"
#include "kernels.h"

namespace vptq {

constexpr int kXThreads = 256;
constexpr int kXE = 4;  // elements per lane and piece

struct GatherXParams {
  const uint32_t* idx;    // [C, N, row_words]
  const char* cent;       // [C, k, V] 2 V bytes per entry
  const char* rcent;      // [C, kr, V] or null
  const uint16_t* x;      // [tokens, I]
  void* y;                // [tokens, O] (dtype, or float when out_f32)
  const uint16_t* scale;  // [I] in column order (perm: scale_permuted) or null (no norm)
  const uint16_t* wbias;  // [I] or null
  const uint16_t* bias;   // [O] or null
  const uint16_t* perm;   // [I] or null
  const uint16_t* oidx;   // [N, S] outlier indices (uint16) or null
  const char* ocent;      // [ko, ov] outlier codebook or null
  int N, G, C, I, O, S, row_words, k, kr, ib, rb, tokens, out_f32;
  int ov, M;              // outlier vector length (V or 4) and rows of oidx (= ceil(O / ov))
  int res_lds;            // the residual table (<= 32 KiB) is copied into LDS and gathered from there
};

constexpr int kXResLdsMax = 32768;

typedef uint32_t u32_a4 __attribute__((aligned(4)));
typedef uint32_t u32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));
typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));

// one codebook entry (V halves = V / 2 words): 16-byte loads; the 24-byte entries of vector
// length 12 are only 8-byte aligned: one 16-byte load at that alignment + one 8-byte load (three
// 8-byte loads were slower than the generic kernel: the gathers are bound by lane addresses per
// clock, not by bytes)
typedef uint32_t u32x4_a8 __attribute__((ext_vector_type(4), aligned(8)));
typedef uint32_t u32x2_a8 __attribute__((ext_vector_type(2), aligned(8)));
typedef uint32_t u32x3_a4 __attribute__((ext_vector_type(3), aligned(4)));
template <int V>
static __device__ __forceinline__ void load_entry(uint32_t (&w)[V / 2], const char* p) {
  if constexpr (V == 12) {
    const u32x4_a8 a = *(const u32x4_a8*)p;
    const u32x2_a8 b = *(const u32x2_a8*)(p + 16);
    w[0] = a[0]; w[1] = a[1]; w[2] = a[2]; w[3] = a[3]; w[4] = b[0]; w[5] = b[1];
  } else if constexpr (V == 10) {   // 20-byte entries, 4-byte aligned
    const u32x4_a4 a = *(const u32x4_a4*)p;
    w[0] = a[0]; w[1] = a[1]; w[2] = a[2]; w[3] = a[3]; w[4] = *(const u32_a4*)(p + 16);
  } else if constexpr (V == 6) {    // 12-byte entries: one global_load_dwordx3
    const u32x3_a4 a = *(const u32x3_a4*)p;
    w[0] = a[0]; w[1] = a[1]; w[2] = a[2];
  } else if constexpr (V == 4) {
    const u32x2_a8 a = *(const u32x2_a8*)p;
    w[0] = a[0]; w[1] = a[1];
  } else if constexpr (V == 2) {
    w[0] = *(const u32_a4*)p;
  } else {
#pragma unroll
    for (int q = 0; q < V / 8; ++q) {
      const u32x4 a = *(const u32x4*)(p + q * 16);
      w[4 * q] = a[0]; w[4 * q + 1] = a[1]; w[4 * q + 2] = a[2]; w[4 * q + 3] = a[3];
    }
  }
}

// the same entry out of the LDS copy of a table (byte offset inside the dynamic LDS segment)
template <int V>
static __device__ __forceinline__ void load_entry_lds(uint32_t (&w)[V / 2], const unsigned char* tab, uint32_t off) {
  typedef __attribute__((address_space(3))) const uint32_t lds_u32_t;
  lds_u32_t* p = (lds_u32_t*)(tab + off);
#pragma unroll
  for (int i = 0; i < V / 2; ++i) w[i] = p[i];   // merged into ds_read_b128 / b64 by the compiler
}

// element at the wave-uniform bit position p of the normalised window n[0..3] (T <= 32 bits)
static __device__ __forceinline__ uint32_t window_elem(const uint32_t (&n)[4], int p, int T) {
  const int wi = p >> 5, s = p & 31;
  const uint32_t lo = wi == 0 ? n[0] : wi == 1 ? n[1] : wi == 2 ? n[2] : n[3];
  const uint32_t hi = wi == 0 ? n[1] : wi == 1 ? n[2] : wi == 2 ? n[3] : 0u;
  const uint32_t mask = T >= 32 ? 0xffffffffu : ((1u << T) - 1u);
  return __builtin_amdgcn_alignbit(hi, lo, (uint32_t)s) & mask;
}

template <typename DT, int V, int TOK, bool PERM>
__global__ __launch_bounds__(kXThreads) void gemv_gatherx_kernel(const GatherXParams P) {
  static_assert(TOK == 1 || TOK == 2 || TOK == 4 || (TOK == 8 && V <= 8), "token slots");
  static_assert(V == 2 || V == 4 || V == 6 || V == 8 || V == 10 || V == 12 || V == 16, "vector length");
  constexpr int E = kXE;
  constexpr int VW = V / 2;             // 32-bit words per entry
  constexpr int VP = V <= 8 ? 8 : 16;   // accumulators padded to a power of two >= 8 (wave reduction)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int row = blockIdx.x;
  const int G = P.G, N = P.N, O = P.O, tokens = P.tokens;
  const int T = P.ib + P.rb;
  const uint32_t mmask = (1u << P.ib) - 1u;
  const bool has_res = P.kr > 0, has_norm = P.scale != nullptr;
  // window offsets are multiples of gcd(4 T, 32): the largest one + 4 T bits must fit 4 words
  const int g32 = (4 * T) & -(4 * T) & 31 ? ((4 * T) & -(4 * T)) : 32;
  const bool need5 = 4 * T > 96 + g32;

  float acc[TOK][V];
#pragma unroll
  for (int t = 0; t < TOK; ++t)
#pragma unroll
    for (int i = 0; i < V; ++i) acc[t][i] = 0.f;
  // small residual tables (<= 32 KiB: kr <= 2048 / 1024 / 1024 entries of 8 / 12 / 16 halves) are
  // copied into LDS per workgroup and codebook group: the kernel is bound by lane addresses per clock
  // in the vector-memory pipe, the copy costs kr of them, the gathers it replaces G each
  extern __shared__ __attribute__((aligned(16))) unsigned char xsmem[];
  const bool res_lds = P.res_lds != 0;

  // ---- outlier columns (the first S input columns, their own codebook; reference
  // quant_gemv.cuh:52-86): one column per lane and step - they are 1-3 % of the layer.  The codebook
  // has either the layer's vector length (one entry per column) or vector length 4 (V / 4 entries of
  // 8 bytes out of V / 4 consecutive rows of the outlier indices).
  for (int c = tid; c < P.S; c += kXThreads) {
    uint32_t w2[VW];
    if (P.ov == V) {
      const uint32_t oi = as_global(P.oidx)[(size_t)row * P.S + c];
      load_entry<V>(w2, as_global(P.ocent) + (size_t)oi * (V * 2));
    } else if constexpr (V % 4 == 0 && V > 4) {
#pragma unroll
      for (int r = 0; r < V / 4; ++r) {
        // rows past the last outlier row belong to outputs >= O, which are not stored: any value does
        const int m = row * (V / 4) + r;
        const uint32_t oi = as_global(P.oidx)[(size_t)(m < P.M ? m : P.M - 1) * P.S + c];
        const u32x2_a8 e2 = *(const u32x2_a8*)(as_global(P.ocent) + (size_t)oi * 8);
        w2[2 * r] = e2[0]; w2[2 * r + 1] = e2[1];
      }
    }
    if (has_norm) {
      const uint32_t s2 = splat16(as_global(P.scale)[c]), b2 = splat16(as_global(P.wbias)[c]);
#pragma unroll
      for (int p = 0; p < VW; ++p) w2[p] = DT::add2(DT::mul2(w2[p], s2), b2);
    }
    const int j = PERM ? (int)as_global(P.perm)[c] : c;
#pragma unroll
    for (int t = 0; t < TOK; ++t) {
      const float xf = DT::to_float(as_global(P.x)[(size_t)(t < tokens ? t : tokens - 1) * P.I + j]);
#pragma unroll
      for (int p = 0; p < VW; ++p) {
        acc[t][2 * p] = DT::fma_lo(w2[p], xf, acc[t][2 * p]);
        acc[t][2 * p + 1] = DT::fma_hi(w2[p], xf, acc[t][2 * p + 1]);
      }
    }
  }

  for (int cb = 0; cb < P.C; ++cb) {
    const uint32_t* const rowp = as_global(P.idx + ((size_t)cb * N + row) * P.row_words);
    const char* const centb = as_global(P.cent + (size_t)cb * P.k * (V * 2));
    const char* const rcentb = has_res ? as_global(P.rcent + (size_t)cb * P.kr * (V * 2)) : nullptr;
    if (res_lds) {
      if (cb > 0) __syncthreads();   // the previous group's table is no longer read
      const int n16 = P.kr * (V * 2) / 16;
      for (int i = tid; i < n16; i += kXThreads)
        *(__attribute__((address_space(3))) u32x4*)(xsmem + (size_t)i * 16) = *(const u32x4*)(rcentb + (size_t)i * 16);
      __syncthreads();
    }
    for (int base = 0; base < G; base += kXThreads * E) {
      const int want = base + tid * E;
      const bool valid = want < G;  // G % 4 == 0 (host check): whole pieces
      const int col0 = valid ? want : G - E;
      const int col = P.S + cb * G + col0;  // position in scale / bias / x
      // per-column scale / bias / activations: E halves each
      uint32_t sp[E / 2] = {0, 0}, bp[E / 2] = {0, 0}, xp[TOK][E / 2];
      if (has_norm) {
        const u32x2_a4 sv = *(const u32x2_a4*)as_global(P.scale + col);
        const u32x2_a4 bv = *(const u32x2_a4*)as_global(P.wbias + col);
        sp[0] = sv[0]; sp[1] = sv[1]; bp[0] = bv[0]; bp[1] = bv[1];
      }
#pragma unroll
      for (int t = 0; t < TOK; ++t) {
        const uint16_t* xr = as_global(P.x + (size_t)(t < tokens ? t : tokens - 1) * P.I);
        const uint32_t keep = valid ? 0xffffffffu : 0u;
        if (PERM) {
          const u32x2_a4 pv = *(const u32x2_a4*)as_global(P.perm + col);
#pragma unroll
          for (int q = 0; q < E / 2; ++q)
            xp[t][q] = ((uint32_t)xr[pv[q] & 0xffffu] | ((uint32_t)xr[pv[q] >> 16] << 16)) & keep;
        } else {
          const u32x2_a4 xv = *(const u32x2_a4*)as_global(xr + col);
          xp[t][0] = xv[0] & keep; xp[t][1] = xv[1] & keep;
        }
      }
      // index window of the lane's 4 elements
      uint32_t n[4];
      {
        const uint32_t bit = (uint32_t)col0 * (uint32_t)T;
        const int w0 = (int)(bit >> 5);
        const uint32_t off = bit & 31u;
        uint32_t w[5];
        const int last = P.row_words - 1;
        if (w0 + 4 <= last) {
          // one 16-byte load at 4-byte alignment (+ the fifth word where some lane's window
          // reaches it: 4 T + offset > 128, a property of T alone): the gathers of this kernel are
          // bound by lane addresses per clock, five 4-byte loads per piece were a fifth of them
          const u32x4_a4 a = *(const u32x4_a4*)(rowp + w0);
          w[0] = a[0]; w[1] = a[1]; w[2] = a[2]; w[3] = a[3];
          w[4] = need5 ? *(const u32_a4*)(rowp + w0 + 4) : 0u;
        } else {
#pragma unroll
          for (int i = 0; i < 5; ++i) w[i] = *(const u32_a4*)(rowp + (w0 + i < last ? w0 + i : last));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) n[i] = __builtin_amdgcn_alignbit(w[i + 1], w[i], off);
      }
      uint32_t mi[E], ri[E];
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const uint32_t v = e == 0 ? (n[0] & (T >= 32 ? 0xffffffffu : ((1u << T) - 1u))) : window_elem(n, e * T, T);
        mi[e] = (v & mmask) * (uint32_t)(V * 2);
        ri[e] = (v >> P.ib) * (uint32_t)(V * 2);
      }
      // all gathers of the piece first (V = 16: two elements at a time), then the arithmetic
      constexpr int GB = V <= 8 ? 4 : 2;  // (4 for the wider entries too: no gain, 64 more registers; 512-thread
                                          // workgroups for the 512 vector-rows of a v = 16 layer of 8192 outputs -
                                          // twice the waves per CU - measured no gain either: the L2 / Infinity Cache
                                          // path of the gathers, not the occupancy, sets the pace)
#pragma unroll
      for (int e0 = 0; e0 < E; e0 += GB) {
        uint32_t cv[GB][VW], rv[GB][VW];
#pragma unroll
        for (int u = 0; u < GB; ++u) {
          load_entry<V>(cv[u], centb + mi[e0 + u]);
          if (res_lds) load_entry_lds<V>(rv[u], xsmem, ri[e0 + u]);
          else if (has_res) load_entry<V>(rv[u], rcentb + ri[e0 + u]);
        }
#pragma unroll
        for (int u = 0; u < GB; ++u) {
          const int e = e0 + u;
          uint32_t w2[VW];
#pragma unroll
          for (int p = 0; p < VW; ++p) w2[p] = has_res ? DT::add2(cv[u][p], rv[u][p]) : cv[u][p];
          if (has_norm) {
#pragma unroll
            for (int p = 0; p < VW; ++p) w2[p] = DT::mul2_bcast(w2[p], sp[e >> 1], e & 1);
#pragma unroll
            for (int p = 0; p < VW; ++p) w2[p] = DT::add2_bcast(w2[p], bp[e >> 1], e & 1);
          }
#pragma unroll
          for (int t = 0; t < TOK; ++t)
#pragma unroll
            for (int p = 0; p < VW; ++p) {
              acc[t][2 * p] = DT::fma_lo_h(w2[p], xp[t][e >> 1], e & 1, acc[t][2 * p]);
              acc[t][2 * p + 1] = DT::fma_hi_h(w2[p], xp[t][e >> 1], e & 1, acc[t][2 * p + 1]);
            }
        }
      }
    }
  }

  constexpr int kVals = TOK * VP;
  __shared__ float red[kXThreads / 64][kVals];
  {
    float v[kVals];
#pragma unroll
    for (int t = 0; t < TOK; ++t)
#pragma unroll
      for (int i = 0; i < VP; ++i) v[t * VP + i] = i < V ? acc[t][i < V ? i : 0] : 0.f;
    using WR = WaveReduce<kVals>;
    WR::run(v, lane);
    if ((lane & ((1 << WR::kShift) - 1)) == 0) red[wave][lane >> WR::kShift] = v[0];
  }
  __syncthreads();
  if (tid < kVals) {
    const int t = tid / VP, i = tid - t * VP;
    const int o = row * V + i;
    if (t < tokens && i < V && o < O) {
      float sum = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
      if (P.bias) sum += DT::to_float(as_global(P.bias)[o]);
      if (P.out_f32) ((float*)as_global((char*)P.y))[(size_t)t * O + o] = sum;
      else ((uint16_t*)as_global((char*)P.y))[(size_t)t * O + o] = DT::from_float(sum);
    }
  }
}

// ---- host side -------------------------------------------------------------------
// token slots of one launch: 8 for the short vectors (<= 64 accumulators), 4 for v = 10 / 12 / 16
int gemv_gatherx_max_chunk(const VptqLayerDesc& d) { return d.vector_len <= 8 ? 8 : 4; }

// codebook entries are 2 v bytes: the alignment the entry loads of load_entry<v> assume
static int entry_align(int v) { return v == 8 || v == 16 ? 16 : (v == 4 || v == 12) ? 8 : 4; }

bool gemv_gatherx_eligible(const VptqLayerDesc& d, int tokens) {
  const int v = d.vector_len;
  const bool norm = d.weight_scale != nullptr && d.weight_bias != nullptr;
  const bool outl = d.outlier_size > 0;
  const bool vlen = v == 2 || v == 4 || v == 6 || v == 8 || v == 10 || v == 12 || v == 16;
  if (!vlen) return false;
  if (outl) {
    const int ov = d.outlier_vector_len;
    const bool same = ov == v && d.num_outlier_indices == d.num_indices;
    const bool four = ov == 4 && v > 4 && (v % 4) == 0 && d.num_outlier_indices >= 1 &&
                      (long long)d.num_outlier_indices * 4 >= d.out_features;
    if (!(same || four) || (d.outlier_size % kXE) != 0 || !d.outlier_indices || !d.outlier_centroids ||
        (((uintptr_t)d.outlier_centroids) & (entry_align(ov) - 1)) != 0 || (((uintptr_t)d.outlier_indices) & 1) != 0)
      return false;
  }
  return d.num_codebooks >= 1 && (d.group_size % kXE) == 0 &&
         d.in_features == d.outlier_size + d.num_codebooks * d.group_size && d.index_bits + d.res_bits <= 32 &&
         (long long)d.row_words * 32 >= (long long)d.group_size * (d.index_bits + d.res_bits) &&
         d.num_indices * v >= d.out_features && tokens >= 1 && tokens <= gemv_gatherx_max_chunk(d) &&
         (d.perm == nullptr || !norm || (d.scale_permuted != nullptr && d.bias_permuted != nullptr)) &&
         (((uintptr_t)d.centroids | (uintptr_t)d.res_centroids) & (entry_align(v) - 1)) == 0 &&
         (((uintptr_t)d.indices | (uintptr_t)d.weight_scale | (uintptr_t)d.weight_bias |
           (uintptr_t)d.scale_permuted | (uintptr_t)d.bias_permuted | (uintptr_t)d.perm) & 3) == 0;
}

template <typename DT, int V, int TOK>
static hipError_t launch_x(const GatherXParams& P, bool perm, hipStream_t st) {
  const dim3 grid(P.N), block(kXThreads);
  const int lds = P.res_lds ? P.kr * (V * 2) : 0;
  if (perm) hipLaunchKernelGGL((gemv_gatherx_kernel<DT, V, TOK, true>), grid, block, lds, st, P);
  else hipLaunchKernelGGL((gemv_gatherx_kernel<DT, V, TOK, false>), grid, block, lds, st, P);
  return hipGetLastError();
}

template <typename DT, int V>
static hipError_t launch_xv(const GatherXParams& P, bool perm, hipStream_t st) {
  const int tok = P.tokens > 4 ? 8 : P.tokens > 2 ? 4 : P.tokens;
  if (tok == 1) return launch_x<DT, V, 1>(P, perm, st);
  if (tok == 2) return launch_x<DT, V, 2>(P, perm, st);
  if (tok == 4) return launch_x<DT, V, 4>(P, perm, st);
  if constexpr (V <= 8) return launch_x<DT, V, 8>(P, perm, st);
  return hipErrorInvalidValue;
}

template <typename DT>
static hipError_t launch_xdt(const GatherXParams& P, int v, bool perm, hipStream_t st) {
  switch (v) {
    case 2: return launch_xv<DT, 2>(P, perm, st);
    case 4: return launch_xv<DT, 4>(P, perm, st);
    case 6: return launch_xv<DT, 6>(P, perm, st);
    case 8: return launch_xv<DT, 8>(P, perm, st);
    case 10: return launch_xv<DT, 10>(P, perm, st);
    case 12: return launch_xv<DT, 12>(P, perm, st);
    case 16: return launch_xv<DT, 16>(P, perm, st);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_gemv_gatherx(const VptqLayerDesc& d, const void* x, void* y, int tokens, bool out_f32,
                               hipStream_t st) {
  GatherXParams P;
  const bool norm = d.weight_scale != nullptr && d.weight_bias != nullptr;
  P.idx = (const uint32_t*)d.indices;
  P.cent = (const char*)d.centroids;
  P.rcent = d.num_res_centroids > 0 ? (const char*)d.res_centroids : nullptr;
  P.x = (const uint16_t*)x;
  P.y = y;
  P.scale = norm ? (const uint16_t*)(d.perm ? d.scale_permuted : d.weight_scale) : nullptr;
  P.wbias = norm ? (const uint16_t*)(d.perm ? d.bias_permuted : d.weight_bias) : nullptr;
  P.bias = (const uint16_t*)d.bias;
  P.perm = d.perm;
  P.oidx = d.outlier_size > 0 ? (const uint16_t*)d.outlier_indices : nullptr;
  P.ocent = d.outlier_size > 0 ? (const char*)d.outlier_centroids : nullptr;
  P.S = d.outlier_size;
  P.ov = d.outlier_size > 0 ? d.outlier_vector_len : 0;
  P.M = d.outlier_size > 0 ? d.num_outlier_indices : 0;
  P.N = d.num_indices; P.G = d.group_size; P.C = d.num_codebooks; P.I = d.in_features; P.O = d.out_features;
  P.row_words = d.row_words;
  P.k = d.num_centroids; P.kr = d.num_res_centroids; P.ib = d.index_bits; P.rb = d.res_bits;
  P.tokens = tokens;
  P.out_f32 = out_f32 ? 1 : 0;
  // (12-half entries are read with 16 + 8 bytes from LDS as well: the table must hold whole 16-byte units)
  const int res_bytes = d.num_res_centroids * d.vector_len * 2;
  P.res_lds = (res_bytes > 0 && res_bytes <= kXResLdsMax && (res_bytes % 16) == 0) ? 1 : 0;
  const bool perm = d.perm != nullptr;
  const int v = d.vector_len;
  return d.dtype == VPTQ_DTYPE_F16 ? launch_xdt<F16>(P, v, perm, st) : launch_xdt<BF16>(P, v, perm, st);
}

}  // namespace vptq
