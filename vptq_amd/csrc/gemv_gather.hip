// Fused dequant + GEMV for the large-codebook VPTQ formats that dominate the published
// checkpoints (v = 8, k = 65536 main centroids; residual: none, 256 or 65536):
//   T = 16  "v8-k65536-0"      T = 24  "v8-k65536-256"      T = 32  "v8-k65536-65536"
// one codebook, no outlier columns, weight_scale/weight_bias present.
//
// Replaces WqA16WithOutliers_PackIndice for those template cases
// (reference csrc/kernels/quant_gemv.cuh:11-186, dispatch csrc/quant_gemv.cu:54-132).
//
// A 1 MiB codebook cannot live in LDS (160 KiB), so centroid rows are gathered from
// L2 / L1 with 16-byte loads, exactly one per index (the residual table, 4 KiB when
// kr = 256, stays L1-resident).  Same decomposition as gemv_k256: a workgroup owns
// ROWS complete vector-rows over all input columns (no split-K, no second kernel),
// lanes stream the packed index rows with wide coalesced loads, all gathers of a
// row-piece are in flight before the first is used, weights are rebuilt with the
// reference's roundings and accumulated in fp32, lane-swap reduce-scatter at the end.
// No LDS tables => occupancy is limited by registers only (8 waves / SIMD), which is
// what hides the gather latency.  Bound: the texture-addresser rate of random 16-byte
// gathers (one lane address per clock), not HBM.
#include "common.h"
#include "kernels.h"

namespace vptq {

constexpr int kGThreads = 256;

struct GatherParams {
  const uint32_t* idx;    // [N, row_words]
  const char* cent;       // [k, 8] 16 bytes per entry
  const char* rcent;      // [kr, 8] or null
  const uint16_t* x;      // [tokens, I]
  uint16_t* y;            // [tokens, O]
  const uint16_t* scale;  // [I] (in column order when perm != null)
  const uint16_t* wbias;  // [I]
  const uint16_t* bias;   // [O] or null
  const uint16_t* perm;   // [I] or null
  int N, G, O, row_words, tokens, out_f32;
};

// Format traits: NW 32-bit words per lane hold E elements of T bits.
// (WIDE: 8 elements of 24 bits = 6 words per lane and piece instead of 4 in 3: the per-column
// scale / bias / x loads and the index load are amortised over twice the gathers - one token,
// wide rows only; with two or more token rows in registers, or few pieces per row, it loses.)
template <int T, bool WIDE = false> struct Fmt;
template <bool W> struct Fmt<16, W> { static constexpr int NW = 4, E = 8; };
template <bool W> struct Fmt<24, W> { static constexpr int E = W ? 8 : 4, NW = E * 3 / 4; };
template <bool W> struct Fmt<32, W> { static constexpr int NW = 4, E = 4; };

template <int T, int NW>
static __device__ __forceinline__ uint32_t elem(const uint32_t (&w)[NW], int e) {
  if (T == 16) return (e & 1) ? (w[e >> 1] >> 16) : (w[e >> 1] & 0xffffu);
  if (T == 32) return w[e];
  // T == 24: 4 elements in 3 words
  const int b = (e >> 2) * 3;
  switch (e & 3) {
    case 0: return w[b] & 0xffffffu;
    case 1: return __builtin_amdgcn_alignbit(w[b + 1], w[b], 24) & 0xffffffu;
    case 2: return __builtin_amdgcn_alignbit(w[b + 2], w[b + 1], 16) & 0xffffffu;
    default: return w[b + 2] >> 8;
  }
}

// A/B knob (tools/gpu_r2_gather.sh): 1 = centroid gathers bypass the vector L1 (nt), 2 = the index
// words too.  Measured: 79 / 63 us per 8192^2 layer against 39.6 us with plain loads - the 9.5 % of
// gathers that hit in L1 and the L1's request merging matter; off.
#ifndef VPTQ_GATHER_NT
#define VPTQ_GATHER_NT 0
#endif

#ifndef VPTQ_GATHER_RES_LDS
#define VPTQ_GATHER_RES_LDS 1
#endif

template <typename DT, int T, int ROWS, int TOK, bool PERM, bool WIDE = false>
__global__ __launch_bounds__(kGThreads) void gemv_gather_kernel(const GatherParams P) {
  using F = Fmt<T, WIDE>;
  constexpr int E = F::E, NW = F::NW;
  constexpr bool RES = T > 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row0 = blockIdx.x * ROWS;
  const int G = P.G, N = P.N, O = P.O, tokens = P.tokens;

  float acc[TOK][ROWS][8];
#pragma unroll
  for (int t = 0; t < TOK; ++t)
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[t][r][i] = 0.f;

  // T = 24 ("v8-k65536-256"): the 256-entry residual table (4 KiB) is copied into LDS once per
  // workgroup and gathered from there.  These kernels are bound by lane ADDRESSES per clock in the
  // vector-memory pipe (one per CU and cycle): the residual gathers were a third of them.
  constexpr bool kResLds = T == 24 && VPTQ_GATHER_RES_LDS;
  __shared__ u32x4 rtab[kResLds ? 256 : 1];
  if constexpr (kResLds) {
    rtab[tid] = *(const u32x4*)(P.rcent + (size_t)tid * 16);   // kGThreads == 256 entries
    __syncthreads();
  }

  for (int base = 0; base < G; base += kGThreads * E) {
    const int want = base + tid * E;
    const bool valid = want < G;  // G % 8 == 0 (host check) => whole pieces
    const int col0 = valid ? want : G - E;
    // per-column scale / bias / activations, E halves each (pairs in 32-bit registers)
    uint32_t sp[E / 2], bp[E / 2], xp[TOK][E / 2];
    {
      const uint32_t* s32 = (const uint32_t*)(P.scale + col0);
      const uint32_t* b32 = (const uint32_t*)(P.wbias + col0);
#pragma unroll
      for (int q = 0; q < E / 2; ++q) { sp[q] = s32[q]; bp[q] = b32[q]; }
#pragma unroll
      for (int t = 0; t < TOK; ++t) {
        const int te = t < tokens ? t : tokens - 1;
        const uint16_t* xr = P.x + (size_t)te * G;
        const uint32_t keep = valid ? 0xffffffffu : 0u;
        if (PERM) {
          const uint32_t* p32 = (const uint32_t*)(P.perm + col0);
#pragma unroll
          for (int q = 0; q < E / 2; ++q) {
            const uint32_t pv = p32[q];
            xp[t][q] = ((uint32_t)xr[pv & 0xffffu] | ((uint32_t)xr[pv >> 16] << 16)) & keep;
          }
        } else {
          const uint32_t* x32 = (const uint32_t*)(xr + col0);
#pragma unroll
          for (int q = 0; q < E / 2; ++q) xp[t][q] = x32[q] & keep;
        }
      }
    }
    uint32_t w[ROWS][NW];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      const int row = row0 + r < N ? row0 + r : N - 1;
      const uint32_t* src = P.idx + (size_t)row * P.row_words + (size_t)col0 * T / 32;
#pragma unroll
      for (int q = 0; q < NW; ++q) w[r][q] = VPTQ_GATHER_NT == 2 ? __builtin_nontemporal_load(src + q) : src[q];
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      // all gathers of this row-piece first, then the arithmetic
      u32x4 cv[E], rv[E];
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const uint32_t v = elem<T, NW>(w[r], e);
        if (VPTQ_GATHER_NT) cv[e] = __builtin_nontemporal_load((const u32x4*)(P.cent + (size_t)(v & 0xffffu) * 16));
        else cv[e] = *(const u32x4*)(P.cent + (size_t)(v & 0xffffu) * 16);
        if constexpr (kResLds) rv[e] = rtab[(v >> 16) & 0xffu];
        else if (RES) rv[e] = *(const u32x4*)(P.rcent + (size_t)(T == 24 ? (v >> 16) & 0xffu : v >> 16) * 16);
      }
#pragma unroll
      for (int e = 0; e < E; ++e) {
        uint32_t w2[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) w2[p] = RES ? DT::add2(cv[e][p], rv[e][p]) : cv[e][p];
#pragma unroll
        for (int p = 0; p < 4; ++p) w2[p] = DT::mul2_bcast(w2[p], sp[e >> 1], e & 1);
#pragma unroll
        for (int p = 0; p < 4; ++p) w2[p] = DT::add2_bcast(w2[p], bp[e >> 1], e & 1);
#pragma unroll
        for (int t = 0; t < TOK; ++t)
#pragma unroll
          for (int p = 0; p < 4; ++p) {
            acc[t][r][2 * p] = DT::fma_lo_h(w2[p], xp[t][e >> 1], e & 1, acc[t][r][2 * p]);
            acc[t][r][2 * p + 1] = DT::fma_hi_h(w2[p], xp[t][e >> 1], e & 1, acc[t][r][2 * p + 1]);
          }
      }
    }
  }

  constexpr int kVals = TOK * ROWS * 8;
  __shared__ float red[kGThreads / 64][kVals];
  {
    float v[kVals];
#pragma unroll
    for (int t = 0; t < TOK; ++t)
#pragma unroll
      for (int r = 0; r < ROWS; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) v[(t * ROWS + r) * 8 + i] = acc[t][r][i];
    using WR = WaveReduce<kVals>;
    WR::run(v, lane);
    if ((lane & ((1 << WR::kShift) - 1)) == 0) red[wave][lane >> WR::kShift] = v[0];
  }
  __syncthreads();
  if (tid < kVals) {
    const int t = tid / (ROWS * 8), rem = tid - t * (ROWS * 8);
    const int row = row0 + (rem >> 3);
    const int o = row * 8 + (rem & 7);
    if (t < tokens && row < N && o < O) {
      float sum = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
      if (P.bias) sum += DT::to_float(P.bias[o]);
      if (P.out_f32) ((float*)P.y)[(size_t)t * O + o] = sum;
      else P.y[(size_t)t * O + o] = DT::from_float(sum);
    }
  }
}

// ---- host side -------------------------------------------------------------------
static int gather_T(const VptqLayerDesc& d) {
  if (d.num_centroids != 65536) return 0;
  if (d.num_res_centroids == 0) return 16;
  if (d.num_res_centroids == 256) return 24;
  if (d.num_res_centroids == 65536) return 32;
  return 0;
}

bool gemv_gather_eligible(const VptqLayerDesc& d, int tokens) {
  const int T = gather_T(d);
  return T != 0 && d.vector_len == 8 && d.num_codebooks == 1 && d.outlier_size == 0 &&
         d.weight_scale != nullptr && d.weight_bias != nullptr && (d.group_size % 8) == 0 &&
         d.group_size == d.in_features && (long long)d.row_words * 32 == (long long)d.group_size * T &&
         tokens >= 1 && tokens <= 8 &&
         (d.perm == nullptr || (d.scale_permuted != nullptr && d.bias_permuted != nullptr)) &&
         (((uintptr_t)d.indices | (uintptr_t)d.centroids | (uintptr_t)d.res_centroids) & 15) == 0 &&
         (((uintptr_t)d.weight_scale | (uintptr_t)d.weight_bias | (uintptr_t)d.scale_permuted |
           (uintptr_t)d.bias_permuted | (uintptr_t)d.perm) & 3) == 0;
}

template <typename DT, int T, int ROWS, int TOK>
static hipError_t launch_rt(const GatherParams& P, bool perm, hipStream_t st) {
  const dim3 grid((P.N + ROWS - 1) / ROWS), block(kGThreads);
  if (perm)
    hipLaunchKernelGGL((gemv_gather_kernel<DT, T, ROWS, TOK, true>), grid, block, 0, st, P);
  else
    hipLaunchKernelGGL((gemv_gather_kernel<DT, T, ROWS, TOK, false>), grid, block, 0, st, P);
  return hipGetLastError();
}

template <typename DT, int T>
static hipError_t launch_t(const GatherParams& P, bool perm, hipStream_t st) {
  // token slots per launch: 1, 2, 4 or 8 (5-8 tokens in ONE pass over the indices and gathers - the
  // gathers, not the FMAs, bound these kernels: two launches of <= 4 tokens cost twice)
  const int tok = P.tokens > 4 ? 8 : P.tokens > 2 ? 4 : P.tokens;
  // ROWS = 2 amortises the per-column scale / bias / x loads when there are enough rows
  if (tok == 1) {
    if constexpr (T == 24) {
      // 8 elements per lane and piece: 8192^2 45.3 -> 42.3 us; loses on short rows (4096: two pieces)
      if (P.G >= 6144) {
        const dim3 grid((P.N + (P.N >= 2048 ? 2 : 1) - 1) / (P.N >= 2048 ? 2 : 1)), block(kGThreads);
        if (P.N >= 2048) {
          if (perm) hipLaunchKernelGGL((gemv_gather_kernel<DT, 24, 2, 1, true, true>), grid, block, 0, st, P);
          else hipLaunchKernelGGL((gemv_gather_kernel<DT, 24, 2, 1, false, true>), grid, block, 0, st, P);
        } else {
          if (perm) hipLaunchKernelGGL((gemv_gather_kernel<DT, 24, 1, 1, true, true>), grid, block, 0, st, P);
          else hipLaunchKernelGGL((gemv_gather_kernel<DT, 24, 1, 1, false, true>), grid, block, 0, st, P);
        }
        return hipGetLastError();
      }
    }
    if (P.N >= 2048) return launch_rt<DT, T, 2, 1>(P, perm, st);
    return launch_rt<DT, T, 1, 1>(P, perm, st);
  }
  if (tok == 2) return launch_rt<DT, T, 1, 2>(P, perm, st);
  if (tok == 4) return launch_rt<DT, T, 1, 4>(P, perm, st);
  return launch_rt<DT, T, 1, 8>(P, perm, st);
}

template <typename DT>
static hipError_t launch_dt(const GatherParams& P, int T, bool perm, hipStream_t st) {
  switch (T) {
    case 16: return launch_t<DT, 16>(P, perm, st);
    case 24: return launch_t<DT, 24>(P, perm, st);
    case 32: return launch_t<DT, 32>(P, perm, st);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_gemv_gather(const VptqLayerDesc& d, const void* x, void* y, int tokens, bool out_f32,
                              hipStream_t st) {
  GatherParams P;
  P.idx = (const uint32_t*)d.indices;
  P.cent = (const char*)d.centroids;
  P.rcent = (const char*)d.res_centroids;
  P.x = (const uint16_t*)x;
  P.y = (uint16_t*)y;
  P.scale = (const uint16_t*)(d.perm ? d.scale_permuted : d.weight_scale);
  P.wbias = (const uint16_t*)(d.perm ? d.bias_permuted : d.weight_bias);
  P.bias = (const uint16_t*)d.bias;
  P.perm = d.perm;
  P.N = d.num_indices;
  P.G = d.group_size;
  P.O = d.out_features;
  P.row_words = d.row_words;
  P.tokens = tokens;
  P.out_f32 = out_f32 ? 1 : 0;
  const int T = gather_T(d);
  return d.dtype == VPTQ_DTYPE_F16 ? launch_dt<F16>(P, T, d.perm != nullptr, st)
                                   : launch_dt<BF16>(P, T, d.perm != nullptr, st);
}

}  // namespace vptq
