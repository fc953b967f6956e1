// Fused dequant + GEMV with LDS-RESIDENT codebooks for the mid-size formats: v = 8, one
// codebook, 256 < k <= 8192 main centroids, 0..512 residual centroids, packed bit stream
// (T = index_bits + res_bits in {12, 13, 20, 21, 22}) or the unpacked v2 wire format
// (uint16 ids [N][I], uint8 / uint16 residual ids).
//
// Replaces, for these formats, WqA16WithOutliers_PackIndice (reference
// csrc/kernels/quant_gemv.cuh:11-186) and ke_quant_gemv_v2 (csrc/kernels/quant_gemv_v2.cuh:
// 15-184; traits csrc/kernels/quant_gemv_traits.cuh:83-214; dispatch csrc/dispatch_macros.h:
// 12-89) - the reference's only TESTED configuration (tests/test_quant_gemv.py:196-219:
// k = 8192, kr = 256 / 512, v = 8).  The reference stages the whole 128 KiB codebook into
// shared memory in EVERY one of its N blocks (N x 132 KiB of L2 -> smem traffic for 2 MiB of
// indices); gemv_generic.hip / gemv_v2.hip gather through L1 / L2 instead (54 us per 8192^2
// layer).  Here:
//  * persistent: one 1024-thread workgroup per CU copies both codebooks into LDS ONCE
//    (k * 16 + kr * 16 bytes <= 136 KiB of the 160 KiB), then walks row groups
//    bid, bid + grid, ...  A row group is RW vector-rows (1..16, chosen by the host so that
//    there are at least as many row groups as CUs); each row is shared by 16 / RW waves,
//    which take interleaved chunks of 512 columns.
//  * lane = 8 consecutive columns of one vector-row: one 16-byte ds_read_b128 gather per
//    codebook and column (un-replicated tables: ~2.5x the conflict-free gather time, still
//    several times faster than an L2 gather), 8 fp32 accumulators per token.
//  * index words: a lane's 8 elements are 8 T bits = a window of <= 7 consecutive 32-bit
//    words starting at a multiple of 8 bits; neighbouring lanes read neighbouring windows
//    (coalesced: every byte of a touched line is used by this or the adjacent instruction).
//    The window is shifted to bit 0 once, then every element sits at a compile-time position.
//    The index stream is read once by one CU: non-temporal loads, so that it does not evict
//    x / scale / bias, which every wave re-reads from L1 / L2 (they are NOT staged in LDS:
//    the codebooks own it).
//  * arithmetic.  fp16: the reference CPU path's roundings, w = r16(r16(r16(c + r) * s) + b)
//    in packed f16 (bit-identical to vptq_dequant), then x * w accumulated in fp32 - the LDS
//    gathers, not the VALU, bound this kernel, so the bit-equivalent form is free here.
//    bf16 (no packed bf16 VALU on gfx950): folded form in fp32,
//    y = sum_g (c + r) * (s_g x_g) + sum_g b_g x_g  (inside the bf16 parity bar; VPTQ_GEMV_EXACT
//    routes bf16 to the generic kernel).
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace vptq {

constexpr int kLThreads = 1024;
constexpr int kLWaves = kLThreads / 64;
constexpr int kLChunk = 64 * 8;           // columns per wave step
constexpr int kLMaxLds = 163840;
constexpr int kFmtV2None = 100, kFmtV2U8 = 101, kFmtV2U16 = 102;  // FMT < 100: packed, T = FMT

struct LdsParams {
  const uint32_t* idx;    // packed: int32 [N][row_words];  v2: uint16 ids [N][G]
  const void* ridx;       // v2: uint8 / uint16 residual ids [N][G] (NULL: none)
  const uint32_t* cent;   // [k][8]
  const uint32_t* rcent;  // [kr][8] or NULL
  const uint16_t* x;      // [tokens][G]
  void* y;                // [tokens][O] (dtype, or float when out_f32)
  const uint16_t* scale;  // [G] in COLUMN order (perm: scale_permuted) or NULL
  const uint16_t* wbias;  // [G] in column order or NULL
  const uint16_t* wbias_plain;  // [G] in input-feature order (sum b * x needs no permutation)
  const uint16_t* bias;   // [O] or NULL
  const uint16_t* perm;   // [G] or NULL
  int N, G, O, row_words, k, kr, ib, rb, tokens, out_f32, RW, n_groups;
};

typedef uint32_t u32a4 __attribute__((aligned(4)));

// ---- index fetch + decode -----------------------------------------------------------
// RAW words a lane holds for its 8 elements
template <int FMT>
struct IdxWindow {
  static constexpr bool kPacked = FMT < 100;
  static constexpr int T = kPacked ? FMT : 16;
  static constexpr bool kAligned = (8 * T) % 32 == 0;            // window starts on a word
  static constexpr int W = kPacked ? (8 * T + (kAligned ? 0 : 24) + 31) / 32 : 4;
  static constexpr int RWORDS = FMT == kFmtV2U8 ? 2 : FMT == kFmtV2U16 ? 4 : 0;
  uint32_t w[W];
  uint32_t r[RWORDS > 0 ? RWORDS : 1];
  uint32_t off;   // packed, unaligned: bit offset of element 0 inside w[0]
  bool slow;      // packed: the window would run past the row - decode element by element
};

template <int FMT>
static __device__ __forceinline__ void fetch_window(IdxWindow<FMT>& q, const LdsParams& P, int row,
                                                    int g0) {
  using Q = IdxWindow<FMT>;
  if constexpr (Q::kPacked) {
    const uint32_t bit = (uint32_t)g0 * (uint32_t)Q::T;
    const uint32_t w0 = bit >> 5;
    q.off = bit & 31u;
    q.slow = (int)w0 + Q::W > P.row_words;
    // a window past the row end is re-read from the start of the row (in bounds) and ignored
    const uint32_t* p = P.idx + (size_t)row * P.row_words + (q.slow ? 0u : w0);
#pragma unroll
    for (int i = 0; i < Q::W; ++i)
      q.w[i] = __builtin_nontemporal_load((const u32a4*)as_global(p + i));
  } else {
    const uint32_t* p = (const uint32_t*)((const uint16_t*)P.idx + (size_t)row * P.G + g0);
#pragma unroll
    for (int i = 0; i < 4; ++i) q.w[i] = __builtin_nontemporal_load(as_global(p + i));
    if constexpr (Q::RWORDS > 0) {
      const size_t e = (size_t)row * P.G + g0;
      const uint32_t* rp = (const uint32_t*)((const char*)P.ridx + e * (FMT == kFmtV2U8 ? 1 : 2));
#pragma unroll
      for (int i = 0; i < Q::RWORDS; ++i) q.r[i] = __builtin_nontemporal_load(as_global(rp + i));
    }
    q.off = 0;
    q.slow = false;
  }
}

// Decoding: normalise the window once (element 0 at bit 0), then element e sits at the
// compile-time position e * T.  -> LDS byte addresses of the main / residual entry.
template <int FMT>
struct IdxDecoded {
  using Q = IdxWindow<FMT>;
  uint32_t w[Q::W];
  uint32_t r[Q::RWORDS > 0 ? Q::RWORDS : 1];
};

template <int FMT>
static __device__ __forceinline__ void normalise_window(IdxDecoded<FMT>& d, const IdxWindow<FMT>& q,
                                                        const LdsParams& P, int row, int g0) {
  using Q = IdxWindow<FMT>;
  if constexpr (Q::kPacked) {
    constexpr int T = Q::T;
    if (q.slow) {
      // the window would cross the row end: rebuild an aligned window element by element
      const uint32_t* rowp = P.idx + (size_t)row * P.row_words;
      uint64_t acc = 0;
      int have = 0, wi = 0;
#pragma unroll
      for (int i = 0; i < Q::W; ++i) d.w[i] = 0;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const uint32_t v = unpack_elem(rowp, g0 + e < P.G ? g0 + e : P.G - 1, T);
        acc |= (uint64_t)v << have;
        have += T;
        if (have >= 32) {
          d.w[wi < Q::W ? wi : Q::W - 1] = (uint32_t)acc;
          acc >>= 32; have -= 32; ++wi;
        }
      }
      if (have > 0 && wi < Q::W) d.w[wi] = (uint32_t)acc;
    } else if constexpr (Q::kAligned) {
#pragma unroll
      for (int i = 0; i < Q::W; ++i) d.w[i] = q.w[i];
    } else {
#pragma unroll
      for (int i = 0; i + 1 < Q::W; ++i) d.w[i] = __builtin_amdgcn_alignbit(q.w[i + 1], q.w[i], q.off);
      d.w[Q::W - 1] = q.w[Q::W - 1] >> q.off;
    }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) d.w[i] = q.w[i];
#pragma unroll
    for (int i = 0; i < (Q::RWORDS > 0 ? Q::RWORDS : 1); ++i) d.r[i] = q.r[i];
  }
}

template <int FMT, int E>
static __device__ __forceinline__ void decode_elem(const IdxDecoded<FMT>& d, const LdsParams& P,
                                                   uint32_t res_base, uint32_t& am, uint32_t& ar) {
  using Q = IdxWindow<FMT>;
  if constexpr (Q::kPacked) {
    constexpr int T = Q::T;
    constexpr uint32_t mask = T >= 32 ? 0xffffffffu : ((1u << T) - 1u);
    constexpr int wi = (T * E) >> 5, sh = (T * E) & 31;
    uint32_t val;
    if constexpr (sh + T <= 32) val = __builtin_amdgcn_ubfe(d.w[wi], (uint32_t)sh, (uint32_t)T);
    else val = __builtin_amdgcn_alignbit(d.w[wi + 1 < Q::W ? wi + 1 : wi], d.w[wi], (uint32_t)sh) & mask;
    am = (val & ((1u << P.ib) - 1u)) << 4;
    ar = res_base + ((val >> P.ib) << 4);
  } else {
    am = __builtin_amdgcn_ubfe(d.w[E >> 1], (uint32_t)(16 * (E & 1)), 16u) << 4;
    if constexpr (FMT == kFmtV2U8)
      ar = res_base + (__builtin_amdgcn_ubfe(d.r[E >> 2], (uint32_t)(8 * (E & 3)), 8u) << 4);
    else if constexpr (FMT == kFmtV2U16)
      ar = res_base + (__builtin_amdgcn_ubfe(d.r[E >> 1], (uint32_t)(16 * (E & 1)), 16u) << 4);
    else
      ar = res_base;
  }
}

// ---- the kernel -----------------------------------------------------------------------
// Main codebook into LDS by LDS-DMA (global_load_lds_dwordx4: 1 KiB per instruction and wave, no registers, no
// ds_write): k x 16 bytes, k a multiple of 64.  Issued as inline assembly (a DMA the compiler can see makes it wait
// for every load in flight); M0 belongs to the compiler: saved and restored inside the statement.  The caller waits
// with s_waitcnt vmcnt(0) before the barrier that publishes the tables.  (Round 3: the copy through registers,
// 2 x 4 loads + 8 ds_write_b128 per thread for k = 8192, kept the first index away for ~4 us per launch.)
#ifndef VPTQ_LDS_DMA
#define VPTQ_LDS_DMA 1
#endif
static __device__ __forceinline__ void lds_dma_table(const uint32_t* cent, int k, int wave, int lane, int n_waves) {
  const int chunks = k >> 6;   // KiB
  const uint64_t base = (uint64_t)(uintptr_t)as_global(cent) + (uint64_t)lane * 16u;
  for (int j = wave; j < chunks; j += n_waves) {
    const uint32_t d = (uint32_t)__builtin_amdgcn_readfirstlane(j * 1024);
    const uint64_t v = base + (uint64_t)j * 1024u;
    uint32_t keep_m0;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep_m0) : "v"(v), "s"(d) : "memory");
  }
}

template <typename DT, int FMT, int TOK>
__global__ __launch_bounds__(kLThreads) void gemv_lds_kernel(const LdsParams P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lsmem[];
  {
    typedef __attribute__((address_space(3))) unsigned char lds_u8_t;
    if ((uint32_t)(uintptr_t)(lds_u8_t*)lsmem != 0u) __builtin_trap();  // absolute LDS addressing
  }
  constexpr bool kF16 = std::is_same<DT, F16>::value;
  constexpr int NV = 8 * TOK;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int G = P.G, N = P.N, O = P.O, tokens = P.tokens;
  const bool has_res = P.kr > 0;

  // LDS map: main table | residual table | red[kLWaves][NV] | bdot[TOK][kLWaves] | bsum[TOK]
  const uint32_t res_base = (uint32_t)P.k * 16u;
  const uint32_t red_off = res_base + (uint32_t)P.kr * 16u;
  float* const red = (float*)(lsmem + red_off);
  float* const bdot_w = red + kLWaves * NV;
  float* const bsum = bdot_w + TOK * kLWaves;

  // ---- prologue: both codebooks into LDS: the main one by LDS-DMA, the residual one (<= 8 KiB, any size) through
  // registers, 4 entries per thread in flight
  {
    const bool dma = VPTQ_LDS_DMA && (P.k & 63) == 0;
    if (dma) lds_dma_table(P.cent, P.k, __builtin_amdgcn_readfirstlane(wave), lane, kLWaves);
    const int total = P.k + P.kr;
    for (int i0 = (dma ? P.k : 0) + tid; i0 < total; i0 += 4 * kLThreads) {
      u32x4 e[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * kLThreads;
        const int ic = i < total ? i : total - 1;
        const uint32_t* src = ic < P.k ? P.cent + (size_t)ic * 4 : P.rcent + (size_t)(ic - P.k) * 4;
        e[u] = *(const u32x4*)as_global(src);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * kLThreads;
        if (i < total) lds_store16((uint32_t)i * 16u, e[u]);
      }
    }
  }
  // bf16 folded form: sum_g b_g x_g once per workgroup (input-feature order: no permutation)
  if constexpr (!kF16) {
    float accb[TOK];
#pragma unroll
    for (int t = 0; t < TOK; ++t) accb[t] = 0.f;
    if (P.wbias_plain) {
      for (int c = tid * 8; c < G; c += kLThreads * 8) {
        const u32x4 bv = *(const u32x4*)as_global(P.wbias_plain + c);
#pragma unroll
        for (int t = 0; t < TOK; ++t) {
          const u32x4 xv = *(const u32x4*)as_global(P.x + (size_t)(t < tokens ? t : tokens - 1) * G + c);
#pragma unroll
          for (int q = 0; q < 4; ++q) accb[t] = DT::dot2(xv[q], bv[q], accb[t]);
        }
      }
    }
#pragma unroll
    for (int t = 0; t < TOK; ++t) {
      const float s = wave_sum(accb[t]);
      if (lane == 0) bdot_w[t * kLWaves + wave] = s;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the LDS-DMA of the codebook is invisible to the compiler)
  __syncthreads();
  if constexpr (!kF16) {
    if (tid < TOK) {
      float s = 0.f;
      for (int w = 0; w < kLWaves; ++w) s += bdot_w[tid * kLWaves + w];
      bsum[tid] = s;
    }
  }

  // ---- row groups
  const int RW = P.RW;            // rows per group (power of two, 1..16)
  const int PW = kLWaves / RW;    // waves per row
  const int rw = wave / PW, part = wave - rw * PW;
  const int n_chunks = (G + kLChunk - 1) / kLChunk;

  for (int rg = blockIdx.x; rg < P.n_groups; rg += gridDim.x) {
    const int row = rg * RW + rw;
    const int rowc = row < N ? row : N - 1;  // spare rows recompute the last one (not stored)
    float acc[TOK][8];
#pragma unroll
    for (int t = 0; t < TOK; ++t)
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[t][i] = 0.f;

    IdxWindow<FMT> nxt;
    if (part < n_chunks) {
      const int g0 = part * kLChunk + lane * 8;
      fetch_window<FMT>(nxt, P, rowc, g0 < G ? g0 : G - 8);
    }
    for (int ch = part; ch < n_chunks; ch += PW) {
      const IdxWindow<FMT> cur = nxt;
      const int g0r = ch * kLChunk + lane * 8;
      const bool valid = g0r < G;
      const int g0 = valid ? g0r : G - 8;  // lanes past G redo the last 8 columns with x = 0
      // activations / scale / bias of the lane's 8 columns (L1 / L2 resident)
      u32x4 xr[TOK], sr = {0, 0, 0, 0}, br = {0, 0, 0, 0};
      if (P.perm) {
        const u32x4 pv = *(const u32x4*)as_global(P.perm + g0);
#pragma unroll
        for (int t = 0; t < TOK; ++t) {
          const uint16_t* xt = as_global(P.x + (size_t)(t < tokens ? t : tokens - 1) * G);
#pragma unroll
          for (int q = 0; q < 4; ++q)
            xr[t][q] = (uint32_t)xt[pv[q] & 0xffffu] | ((uint32_t)xt[pv[q] >> 16] << 16);
        }
      } else {
#pragma unroll
        for (int t = 0; t < TOK; ++t)
          xr[t] = *(const u32x4*)as_global(P.x + (size_t)(t < tokens ? t : tokens - 1) * G + g0);
      }
      if (P.scale) sr = *(const u32x4*)as_global(P.scale + g0);
      if (P.wbias) br = *(const u32x4*)as_global(P.wbias + g0);
      // next chunk's index words
      if (ch + PW < n_chunks) {
        const int gn = (ch + PW) * kLChunk + lane * 8;
        fetch_window<FMT>(nxt, P, rowc, gn < G ? gn : G - 8);
      }
      if (!valid) {
#pragma unroll
        for (int t = 0; t < TOK; ++t) xr[t] = u32x4{0, 0, 0, 0};
      }
      IdxDecoded<FMT> dec;
      normalise_window<FMT>(dec, cur, P, rowc, g0);
      // gathers kGB elements at a time: all 16 gathers of a chunk in flight for one fp16 token
      // (+1...5 % over 8 in flight), fewer with more token accumulators (no spills)
#ifndef VPTQ_LDS_GB1
#define VPTQ_LDS_GB1 8
#endif
      constexpr int kGB = kF16 ? (TOK == 4 ? 2 : TOK == 1 ? VPTQ_LDS_GB1 : 4) : (TOK == 1 ? 4 : 2);
      auto batch = [&](auto e0c) {
        constexpr int e0 = decltype(e0c)::value;
        u32x4 cv[kGB], rv[kGB];
        auto gather1 = [&](auto uc) {
          constexpr int u = decltype(uc)::value;
          uint32_t am, ar;
          decode_elem<FMT, e0 + u>(dec, P, res_base, am, ar);
          cv[u] = lds_load16(am);
          if (has_res) rv[u] = lds_load16(ar);
        };
        gather1(std::integral_constant<int, 0>{});
        if constexpr (kGB > 1) gather1(std::integral_constant<int, 1>{});
        if constexpr (kGB > 2) { gather1(std::integral_constant<int, 2>{}); gather1(std::integral_constant<int, 3>{}); }
        if constexpr (kGB > 4) {
          gather1(std::integral_constant<int, 4>{}); gather1(std::integral_constant<int, 5>{});
          gather1(std::integral_constant<int, 6>{}); gather1(std::integral_constant<int, 7>{});
        }
#pragma unroll
        for (int u = 0; u < kGB; ++u) {
          const int e = e0 + u, q = e >> 1, h = e & 1;
          if constexpr (kF16) {
            uint32_t w2[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) w2[p] = has_res ? DT::add2(cv[u][p], rv[u][p]) : cv[u][p];
            if (P.scale) {
#pragma unroll
              for (int p = 0; p < 4; ++p) w2[p] = DT::mul2_bcast(w2[p], sr[q], h);
            }
            if (P.wbias) {
#pragma unroll
              for (int p = 0; p < 4; ++p) w2[p] = DT::add2_bcast(w2[p], br[q], h);
            }
#pragma unroll
            for (int t = 0; t < TOK; ++t)
#pragma unroll
              for (int p = 0; p < 4; ++p) {
                acc[t][2 * p] = DT::fma_lo_h(w2[p], xr[t][q], h, acc[t][2 * p]);
                acc[t][2 * p + 1] = DT::fma_hi_h(w2[p], xr[t][q], h, acc[t][2 * p + 1]);
              }
          } else {
            const float sf = P.scale ? DT::half_of(sr[q], h) : 1.f;
            float xs[TOK];
#pragma unroll
            for (int t = 0; t < TOK; ++t) xs[t] = sf * DT::half_of(xr[t][q], h);
#pragma unroll
            for (int p = 0; p < 4; ++p) {
              float wl = DT::lo(cv[u][p]), wh = DT::hi(cv[u][p]);
              if (has_res) { wl += DT::lo(rv[u][p]); wh += DT::hi(rv[u][p]); }
#pragma unroll
              for (int t = 0; t < TOK; ++t) {
                acc[t][2 * p] = __builtin_fmaf(wl, xs[t], acc[t][2 * p]);
                acc[t][2 * p + 1] = __builtin_fmaf(wh, xs[t], acc[t][2 * p + 1]);
              }
            }
          }
        }
      };
      batch(std::integral_constant<int, 0>{});
      if constexpr (kGB < 8) batch(std::integral_constant<int, kGB>{});
      if constexpr (kGB < 4) { batch(std::integral_constant<int, 2 * kGB>{}); batch(std::integral_constant<int, 3 * kGB>{}); }
    }
    // ---- reduce: over the lanes of the wave, then over the PW waves of the row
    float v[NV];
#pragma unroll
    for (int t = 0; t < TOK; ++t)
#pragma unroll
      for (int i = 0; i < 8; ++i) v[t * 8 + i] = acc[t][i];
    WaveReduce<NV>::run(v, lane);
    constexpr int kShift = WaveReduce<NV>::kShift;
    if ((lane & ((1 << kShift) - 1)) == 0) red[wave * NV + (lane >> kShift)] = v[0];
    __syncthreads();
    if (tid < RW * NV) {
      const int r2 = tid / NV, ent = tid - r2 * NV;
      const int t = ent >> 3, i = ent & 7;
      const int orow = rg * RW + r2;
      const int o = orow * 8 + i;
      if (t < tokens && orow < N && o < O) {
        float s = 0.f;
        for (int p = 0; p < PW; ++p) s += red[(r2 * PW + p) * NV + ent];
        if constexpr (!kF16) s += bsum[t];
        if (P.bias) s += DT::to_float(as_global(P.bias)[o]);
        if (P.out_f32) ((float*)as_global((uint16_t*)P.y))[(size_t)t * O + o] = s;
        else as_global((uint16_t*)P.y)[(size_t)t * O + o] = DT::from_float(s);
      }
    }
    __syncthreads();
  }
}

// ---- one token, folded form, MFMA accumulate ---------------------------------------------
// The kernel above spends ~37 VALU instructions per index in the reference's roundings (PMC: VALU
// busy 73 % of the launch, LDS 24 %).  One token in the default arithmetic takes the route of
// gemv_k256m.hip instead:
//   y = sum_g (c + r) * f16(s_g x_g) + sum_g b_g x_g + bias        (fp32 accumulation)
// with the multiply-accumulate on the matrix pipe: lane = (column chunk blk = lane >> 2, vector-row
// j = lane & 3); v_mfma_f32_4x4x4 with X = x' * e_j and W = the gathered halves accumulates x' times
// the lane's own four weights (16 independent blocks; gemv_k256m.hip's file comment).  f16(s x) is
// staged in LDS once per workgroup behind the codebooks (2 bytes per column), sum b x is computed
// once.  Per index: the address arithmetic, two operand perms and 4 (no residual: 2) MFMAs.
// Needs row groups of >= 4 vector-rows (N >= 4 x CUs) and LDS for the staged activations; VPTQ_GEMV_EXACT,
// 2-4 tokens and smaller layers keep the kernel above.
constexpr int kLMCols = 128;       // columns per wave step: 16 chunks of 8
constexpr int kLMStageIt = 4;      // staging passes of 8192 columns (G <= 32768)

static int lds_mfma_bytes(int k, int kr, int G) {
  return (k + kr) * 16 + G * 2 + 16 + (kLWaves * 32 + kLWaves + 16) * 4;
}

template <typename DT, int FMT>
__global__ __launch_bounds__(kLThreads) void gemv_lds_mfma_kernel(const LdsParams P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lsmem[];
  {
    typedef __attribute__((address_space(3))) unsigned char lds_u8_t;
    if ((uint32_t)(uintptr_t)(lds_u8_t*)lsmem != 0u) __builtin_trap();  // absolute LDS addressing
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 3, blk = lane >> 2;
  const int G = P.G, N = P.N, O = P.O;
  const bool has_res = P.kr > 0;

  // LDS map: main table | residual table | f16(s x): G halves + 8 zeros | red[kLWaves][32] | bdot[kLWaves] | bsum
  const uint32_t res_base = (uint32_t)P.k * 16u;
  const uint32_t xs_off = res_base + (uint32_t)P.kr * 16u;
  const uint32_t red_off = xs_off + (uint32_t)G * 2u + 16u;
  float* const red = (float*)(lsmem + red_off);
  float* const bdot_w = red + kLWaves * 32;
  float* const bsum = bdot_w + kLWaves;

  // row groups of RW = 4 * sets vector-rows; the 16 / sets waves of a set take interleaved steps of
  // 128 columns
  const int RW = P.RW;
  const int sets = RW >> 2;
  const int PW = kLWaves / sets;
  const int set = wave / PW, part = wave - set * PW;
  const int n_steps = (G + kLMCols - 1) / kLMCols;
  // (requesting the first row group's first index window ahead of the prologue, so that its HBM latency
  // passes while the codebooks are copied, changed nothing: +-0.3 us either way over four formats)
  // ---- prologue: codebooks into LDS (the main one by LDS-DMA, the residual one through registers), activations staged
  {
    const bool dma = VPTQ_LDS_DMA && (P.k & 63) == 0;
    if (dma) lds_dma_table(P.cent, P.k, __builtin_amdgcn_readfirstlane(wave), lane, kLWaves);
    const int total = P.k + P.kr;
    for (int i0 = (dma ? P.k : 0) + tid; i0 < total; i0 += 4 * kLThreads) {
      u32x4 e[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * kLThreads;
        const int ic = i < total ? i : total - 1;
        const uint32_t* src = ic < P.k ? P.cent + (size_t)ic * 4 : P.rcent + (size_t)(ic - P.k) * 4;
        e[u] = *(const u32x4*)as_global(src);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * kLThreads;
        if (i < total) lds_store16((uint32_t)i * 16u, e[u]);
      }
    }
  }
  {
    // sum b x in input-feature order; x' = f16(s x) in COLUMN order (perm: x parked in LDS in its own
    // order first, gathered through the permutation from there - 8192 scattered 2-byte global loads
    // per workgroup cost microseconds)
    float accb = 0.f;
    u32x4 xv[kLMStageIt];
#pragma unroll
    for (int it = 0; it < kLMStageIt; ++it) {
      const int c = (it * kLThreads + tid) * 8;
      if (c < G) {
        xv[it] = *(const u32x4*)as_global(P.x + c);
        if (P.wbias_plain) {
          const u32x4 bv = *(const u32x4*)as_global(P.wbias_plain + c);
#pragma unroll
          for (int q = 0; q < 4; ++q) accb = DT::dot2(xv[it][q], bv[q], accb);
        }
      }
    }
    if (P.perm) {
#pragma unroll
      for (int it = 0; it < kLMStageIt; ++it) {
        const int c = (it * kLThreads + tid) * 8;
        if (c < G) lds_store16(xs_off + (uint32_t)c * 2u, xv[it]);
      }
      __syncthreads();
      typedef __attribute__((address_space(3))) const uint16_t lds_u16_t;
#pragma unroll
      for (int it = 0; it < kLMStageIt; ++it) {
        const int c = (it * kLThreads + tid) * 8;
        if (c < G) {
          const u32x4 pv = *(const u32x4*)as_global(P.perm + c);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const uint32_t lo = *(lds_u16_t*)(uintptr_t)(xs_off + (pv[q] & 0xffffu) * 2u);
            const uint32_t hi = *(lds_u16_t*)(uintptr_t)(xs_off + (pv[q] >> 16) * 2u);
            xv[it][q] = lo | (hi << 16);
          }
        }
      }
      __syncthreads();   // every thread has its activations: the area may be overwritten
    }
#pragma unroll
    for (int it = 0; it < kLMStageIt; ++it) {
      const int c = (it * kLThreads + tid) * 8;
      if (c < G) {
        u32x4 v = xv[it];
        if (P.scale) {
          const u32x4 sv = *(const u32x4*)as_global(P.scale + c);
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] = DT::mul2(v[q], sv[q]);
        }
        lds_store16(xs_off + (uint32_t)c * 2u, v);
      }
    }
    if (tid == 0) lds_store16(xs_off + (uint32_t)G * 2u, u32x4{0, 0, 0, 0});  // the operand of columns past G
    const float sb = wave_sum(accb);
    if (lane == 0) bdot_w[wave] = sb;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the LDS-DMA of the codebook is invisible to the compiler)
  __syncthreads();
  if (tid == 0) {
    float sb = 0.f;
    for (int w = 0; w < kLWaves; ++w) sb += bdot_w[w];
    bsum[0] = sb;
  }

  // x operand of the MFMA: x' * e_j as two packed pairs cut out of a packed x' register
  const uint32_t selA[2] = {j == 0 ? 0x0c0c0504u : j == 1 ? 0x05040c0cu : 0x0c0c0c0cu,
                            j == 0 ? 0x0c0c0706u : j == 1 ? 0x07060c0cu : 0x0c0c0c0cu};
  const uint32_t selB[2] = {j == 2 ? 0x0c0c0504u : j == 3 ? 0x05040c0cu : 0x0c0c0c0cu,
                            j == 2 ? 0x0c0c0706u : j == 3 ? 0x07060c0cu : 0x0c0c0c0cu};

  for (int rg = blockIdx.x; rg < P.n_groups; rg += gridDim.x) {
    const int row = rg * RW + set * 4 + j;
    const int rowc = row < N ? row : N - 1;  // spare rows recompute the last one (not stored)
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};

    // index windows: one step ahead of the arithmetic (two steps ahead - 2 x 21-28 KiB in flight per CU -
    // measured SLOWER, 14.9 vs 13.7 us at k = 8192 + 256: as in gemv_k256m.hip, a deeper request burst at the
    // start of a short kernel delays the first data more than it hides later latency;
    // profiles/r02/gemv_lds_mfma_ab.txt)
#ifndef VPTQ_LDS_MFMA_DEPTH
#define VPTQ_LDS_MFMA_DEPTH 1
#endif
    IdxWindow<FMT> nxt, nxt2;
    auto fetch_step = [&](IdxWindow<FMT>& q, int st) {
      if (st < n_steps) {
        const int g0 = st * kLMCols + blk * 8;
        fetch_window<FMT>(q, P, rowc, g0 < G ? g0 : G - 8);
      }
    };
    fetch_step(nxt, part);
    if (VPTQ_LDS_MFMA_DEPTH > 1) fetch_step(nxt2, part + PW);
    for (int st = part; st < n_steps; st += PW) {
      const IdxWindow<FMT> cur = nxt;
      const int g0r = st * kLMCols + blk * 8;
      const bool valid = g0r < G;
      const int g0 = valid ? g0r : G - 8;     // chunks past G redo the last one against x' = 0
      const u32x4 xq = lds_load16(xs_off + (uint32_t)(valid ? g0r : G) * 2u);
      if (VPTQ_LDS_MFMA_DEPTH > 1) {
        nxt = nxt2;
        fetch_step(nxt2, st + 2 * PW);
      } else {
        fetch_step(nxt, st + PW);
      }
      IdxDecoded<FMT> dec;
      normalise_window<FMT>(dec, cur, P, rowc, g0);
      u32x4 cv[8], rv[8];
      auto gather1 = [&](auto ec) {
        constexpr int e = decltype(ec)::value;
        uint32_t am, ar;
        decode_elem<FMT, e>(dec, P, res_base, am, ar);
        cv[e] = lds_load16(am);
        if (has_res) rv[e] = lds_load16(ar);
      };
      gather1(std::integral_constant<int, 0>{}); gather1(std::integral_constant<int, 1>{});
      gather1(std::integral_constant<int, 2>{}); gather1(std::integral_constant<int, 3>{});
      gather1(std::integral_constant<int, 4>{}); gather1(std::integral_constant<int, 5>{});
      gather1(std::integral_constant<int, 6>{}); gather1(std::integral_constant<int, 7>{});
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int q = e >> 1, h = e & 1;
        const u32x2 xo = {__builtin_amdgcn_perm(xq[q], 0u, selA[h]), __builtin_amdgcn_perm(xq[q], 0u, selB[h])};
        acc0 = DT::mfma4(xo, u32x2{cv[e][0], cv[e][1]}, acc0);
        acc1 = DT::mfma4(xo, u32x2{cv[e][2], cv[e][3]}, acc1);
        if (has_res) {
          acc0 = DT::mfma4(xo, u32x2{rv[e][0], rv[e][1]}, acc0);
          acc1 = DT::mfma4(xo, u32x2{rv[e][2], rv[e][3]}, acc1);
        }
      }
    }
    // ---- reduce over the 16 column chunks of the wave: lane bits 5 and 4 by swap-and-add (halving the
    // values carried), bits 3 and 2 by DPP row rotations, which keep lane & 3; afterwards lane l holds
    // outputs 4 * bit5 + 2 * bit4 + {0, 1} of vector-row l & 3.  Then over the PW waves of the set.
    float v[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] = acc0[i]; v[4 + i] = acc1[i]; }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[i]), __float_as_uint(v[i + 4]), false, false);
      v[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[i]), __float_as_uint(v[i + 2]), false, false);
      v[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) v[i] = row_ror_add<4>(row_ror_add<8>(v[i]));
    if ((lane & 12) == 0) {
      const int t0 = ((lane >> 5) & 1) * 4 + ((lane >> 4) & 1) * 2;
      red[(wave * 4 + j) * 8 + t0] = v[0];
      red[(wave * 4 + j) * 8 + t0 + 1] = v[1];
    }
    __syncthreads();
    if (tid < RW * 8) {
      const int r2 = tid >> 3, i = tid & 7;
      const int orow = rg * RW + r2;
      const int o = orow * 8 + i;
      if (orow < N && o < O) {
        const int st2 = r2 >> 2, j2 = r2 & 3;
        float sum = 0.f;
        for (int p = 0; p < PW; ++p) sum += red[((st2 * PW + p) * 4 + j2) * 8 + i];
        sum += bsum[0];
        if (P.bias) sum += DT::to_float(as_global(P.bias)[o]);
        if (P.out_f32) ((float*)as_global((uint16_t*)P.y))[o] = sum;
        else as_global((uint16_t*)P.y)[o] = DT::from_float(sum);
      }
    }
    __syncthreads();
  }
}

// ---- host side ------------------------------------------------------------------------
static int lds_cus() {
  static std::atomic<int> cus[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (!cus[dev]) {
    hipDeviceProp_t p;
    cus[dev] = hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0
                   ? p.multiProcessorCount : 256;
  }
  return cus[dev];
}

static int lds_rows_per_group(int N) {
  const int cus = lds_cus();
  int RW = kLWaves;
  while (RW > 1 && (N + RW - 1) / RW < cus) RW >>= 1;
  return RW;
}

static int lds_bytes(int k, int kr, int tok) {
  return (k + kr) * 16 + (kLWaves * 8 * tok + tok * kLWaves + tok + 16) * 4;
}

static bool lds_fmt_ok(int fmt) {
  return fmt == 12 || fmt == 13 || fmt == 20 || fmt == 21 || fmt == 22 || fmt == kFmtV2None ||
         fmt == kFmtV2U8 || fmt == kFmtV2U16;
}

template <typename DT, int FMT, int TOK>
static hipError_t launch_lds_t(const LdsParams& P, int grid, int lds, hipStream_t st) {
  auto kern = gemv_lds_kernel<DT, FMT, TOK>;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  static std::atomic<bool> attr_set[64];
  if (!attr_set[dev]) {
    const hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, kLMaxLds);
    if (e != hipSuccess) return e;
    attr_set[dev] = true;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(kLThreads), lds, st, P);
  return hipGetLastError();
}

template <typename DT, int FMT>
static hipError_t launch_lds_fmt(const LdsParams& P, int tok, int grid, int lds, hipStream_t st) {
  if (tok == 1) return launch_lds_t<DT, FMT, 1>(P, grid, lds, st);
  if (tok == 2) return launch_lds_t<DT, FMT, 2>(P, grid, lds, st);
  // four token slots: fp16 only (the widened bf16 arithmetic would spill)
  if constexpr (std::is_same<DT, F16>::value) return launch_lds_t<DT, FMT, 4>(P, grid, lds, st);
  return hipErrorInvalidValue;
}

template <typename DT>
static hipError_t launch_lds_dt(const LdsParams& P, int fmt, int tok, int grid, int lds, hipStream_t st) {
  switch (fmt) {
    case 12: return launch_lds_fmt<DT, 12>(P, tok, grid, lds, st);
    case 13: return launch_lds_fmt<DT, 13>(P, tok, grid, lds, st);
    case 20: return launch_lds_fmt<DT, 20>(P, tok, grid, lds, st);
    case 21: return launch_lds_fmt<DT, 21>(P, tok, grid, lds, st);
    case 22: return launch_lds_fmt<DT, 22>(P, tok, grid, lds, st);
    case kFmtV2None: return launch_lds_fmt<DT, kFmtV2None>(P, tok, grid, lds, st);
    case kFmtV2U8: return launch_lds_fmt<DT, kFmtV2U8>(P, tok, grid, lds, st);
    case kFmtV2U16: return launch_lds_fmt<DT, kFmtV2U16>(P, tok, grid, lds, st);
    default: return hipErrorInvalidValue;
  }
}

template <typename DT, int FMT>
static hipError_t launch_lds_mfma_t(const LdsParams& P, int grid, int lds, hipStream_t st) {
  auto kern = gemv_lds_mfma_kernel<DT, FMT>;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  static std::atomic<bool> attr_set[64];
  if (!attr_set[dev]) {
    const hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, kLMaxLds);
    if (e != hipSuccess) return e;
    attr_set[dev] = true;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(kLThreads), lds, st, P);
  return hipGetLastError();
}

template <typename DT>
static hipError_t launch_lds_mfma_dt(const LdsParams& P, int fmt, int grid, int lds, hipStream_t st) {
  switch (fmt) {
    case 12: return launch_lds_mfma_t<DT, 12>(P, grid, lds, st);
    case 13: return launch_lds_mfma_t<DT, 13>(P, grid, lds, st);
    case 20: return launch_lds_mfma_t<DT, 20>(P, grid, lds, st);
    case 21: return launch_lds_mfma_t<DT, 21>(P, grid, lds, st);
    case 22: return launch_lds_mfma_t<DT, 22>(P, grid, lds, st);
    case kFmtV2None: return launch_lds_mfma_t<DT, kFmtV2None>(P, grid, lds, st);
    case kFmtV2U8: return launch_lds_mfma_t<DT, kFmtV2U8>(P, grid, lds, st);
    case kFmtV2U16: return launch_lds_mfma_t<DT, kFmtV2U16>(P, grid, lds, st);
    default: return hipErrorInvalidValue;
  }
}

// one token in the default arithmetic on a layer with row groups of >= 4 vector-rows: the MFMA kernel
static bool lds_use_mfma(const LdsParams& P, int RW, bool exact) {
  static std::atomic<int> force{-1};   // VPTQ_LDS_KERNEL=valu|mfma (A/B runs)
  if (force < 0) {
    const char* ev = vptq::tune_env("VPTQ_LDS_KERNEL");
    force = ev && ev[0] == 'v' ? 1 : 0;
  }
  return force != 1 && P.tokens == 1 && !exact && RW >= 4 && P.G <= kLMStageIt * kLThreads * 8 &&
         lds_mfma_bytes(P.k, P.kr, P.G) <= kLMaxLds && (((uintptr_t)P.x | (uintptr_t)P.scale | (uintptr_t)P.wbias_plain | (uintptr_t)P.perm) & 15) == 0;
}

// common tail: row-group geometry + launch (tokens <= 4 per launch)
static hipError_t launch_lds(LdsParams& P, int fmt, bool f16, bool exact, hipStream_t st) {
  if (!f16 && P.tokens > 2) return hipErrorInvalidValue;  // (bf16: 2 token slots, see gemv_lds_max_chunk)
  const int tok = P.tokens > 2 ? 4 : P.tokens;
  const int lds = lds_bytes(P.k, P.kr, tok);
  if (lds > kLMaxLds || !lds_fmt_ok(fmt)) return hipErrorInvalidValue;
  const int cus = lds_cus();
  const int RW = lds_rows_per_group(P.N);
  P.RW = RW;
  P.n_groups = (P.N + RW - 1) / RW;
  const int grid = P.n_groups < cus ? P.n_groups : cus;
  if (lds_use_mfma(P, RW, exact)) {
    const int ldsm = lds_mfma_bytes(P.k, P.kr, P.G);
    return f16 ? launch_lds_mfma_dt<F16>(P, fmt, grid, ldsm, st) : launch_lds_mfma_dt<BF16>(P, fmt, grid, ldsm, st);
  }
  return f16 ? launch_lds_dt<F16>(P, fmt, tok, grid, lds, st) : launch_lds_dt<BF16>(P, fmt, tok, grid, lds, st);
}

// tokens one launch takes: 4 (fp16) / 2 (bf16)
int gemv_lds_max_chunk(int dtype) { return dtype == VPTQ_DTYPE_F16 ? 4 : 2; }

bool gemv_lds_eligible(const VptqLayerDesc& d, int tokens, int flags) {
  if (d.vector_len != 8 || d.num_codebooks != 1 || d.outlier_size != 0) return false;
  if (d.num_centroids <= 256 || d.num_centroids > 8192 || d.num_res_centroids > 512) return false;
  if (d.group_size != d.in_features || (d.group_size & 7)) return false;
  if (!lds_fmt_ok(d.index_bits + d.res_bits)) return false;
  if (lds_bytes(d.num_centroids, d.num_res_centroids, 4) > kLMaxLds) return false;
  if ((d.weight_scale != nullptr) && d.perm && !(d.scale_permuted && d.bias_permuted)) return false;
  // bf16 computes the folded form here: the caller asked for the reference's roundings
  if (d.dtype != VPTQ_DTYPE_F16 && (flags & VPTQ_GEMV_EXACT)) return false;
  // every tensor the kernels read with 16-byte loads (scale / bias / their permuted copies / perm; the codebooks)
  if ((((uintptr_t)d.centroids | (uintptr_t)d.res_centroids | (uintptr_t)d.weight_scale | (uintptr_t)d.weight_bias |
        (uintptr_t)d.scale_permuted | (uintptr_t)d.bias_permuted | (uintptr_t)d.perm) & 15) != 0) return false;
  return tokens >= 1 && tokens <= 4;
}

// which of the two kernels a one-call launch of `tokens` tokens takes (host logic, no launch)
const char* gemv_lds_name(const VptqLayerDesc& d, int tokens, int flags) {
  LdsParams P = {};
  P.k = d.num_centroids; P.kr = d.num_res_centroids; P.G = d.group_size; P.tokens = tokens;
  const bool exact = (flags & VPTQ_GEMV_EXACT) != 0 || tokens > 1;
  return lds_use_mfma(P, lds_rows_per_group(d.num_indices), exact) ? "gemv_lds_mfma_kernel" : "gemv_lds_kernel";
}

hipError_t launch_gemv_lds(const VptqLayerDesc& d, const void* x, void* y, int tokens, bool out_f32,
                           int flags, hipStream_t st) {
  LdsParams P = {};
  P.idx = (const uint32_t*)d.indices;
  P.ridx = nullptr;
  P.cent = (const uint32_t*)d.centroids;
  P.rcent = (const uint32_t*)d.res_centroids;
  P.x = (const uint16_t*)x;
  P.y = y;
  const bool norm = d.weight_scale != nullptr;
  P.scale = (const uint16_t*)(norm ? (d.perm ? d.scale_permuted : d.weight_scale) : nullptr);
  P.wbias = (const uint16_t*)(norm ? (d.perm ? d.bias_permuted : d.weight_bias) : nullptr);
  P.wbias_plain = (const uint16_t*)(norm ? d.weight_bias : nullptr);
  P.bias = (const uint16_t*)d.bias;
  P.perm = d.perm;
  P.N = d.num_indices; P.G = d.group_size; P.O = d.out_features; P.row_words = d.row_words;
  P.k = d.num_centroids; P.kr = d.num_res_centroids; P.ib = d.index_bits; P.rb = d.res_bits;
  P.tokens = tokens; P.out_f32 = out_f32 ? 1 : 0;
  return launch_lds(P, d.index_bits + d.res_bits, d.dtype == VPTQ_DTYPE_F16, (flags & VPTQ_GEMV_EXACT) != 0, st);
}

bool gemv_lds_v2_eligible(const VptqV2Desc& d, int tokens) {
  if (d.vector_len != 8 || d.num_centroids < 1 || d.num_centroids > 8192 || d.num_res_centroids > 512)
    return false;
  if ((d.in_features & 7) || lds_bytes(d.num_centroids, d.num_res_centroids, 4) > kLMaxLds) return false;
  if ((((uintptr_t)d.centroids | (uintptr_t)d.res_centroids | (uintptr_t)d.indices) & 15) != 0) return false;
  if (d.num_res_centroids > 0 && (((uintptr_t)d.res_indices) & 7) != 0) return false;
  return tokens >= 1 && tokens <= 4;
}

hipError_t launch_gemv_lds_v2(const VptqV2Desc& d, const void* x, void* y, int tokens, bool out_f32,
                              int flags, hipStream_t st) {
  LdsParams P = {};
  P.idx = (const uint32_t*)d.indices;
  P.ridx = d.num_res_centroids > 0 ? d.res_indices : nullptr;
  P.cent = (const uint32_t*)d.centroids;
  P.rcent = (const uint32_t*)(d.num_res_centroids > 0 ? d.res_centroids : nullptr);
  P.x = (const uint16_t*)x;
  P.y = y;
  P.scale = (const uint16_t*)d.scale_weights;
  P.wbias = (const uint16_t*)d.scale_bias;
  P.wbias_plain = (const uint16_t*)d.scale_bias;
  P.bias = (const uint16_t*)d.bias;
  P.perm = nullptr;
  P.N = d.out_features / 8; P.G = d.in_features; P.O = d.out_features; P.row_words = 0;
  P.k = d.num_centroids; P.kr = d.num_res_centroids > 0 ? d.num_res_centroids : 0;
  P.ib = 16; P.rb = 0;
  P.tokens = tokens; P.out_f32 = out_f32 ? 1 : 0;
  const int fmt = d.num_res_centroids <= 0 ? kFmtV2None : d.res_index_bytes == 1 ? kFmtV2U8 : kFmtV2U16;
  return launch_lds(P, fmt, d.dtype == VPTQ_DTYPE_F16, (flags & VPTQ_GEMV_EXACT) != 0, st);
}

}  // namespace vptq
