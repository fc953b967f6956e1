// Fused dequant + GEMV for 2 - 8 TOKENS over the load-time derived sliced layouts of gemv_sliced.hip (large-codebook
// formats: v = 8 / 16, 16384 ... 65536 main centroids, any residual codebook; reference: WqA16WithOutliers_PackIndice,
// csrc/kernels/quant_gemv.cuh:11-186, which serves these token counts - vptq/ops/quant_gemm.py:213 - by gathering every
// centroid row from the 1 - 2 MiB codebook through the caches once per token batch).
//
// One token fills the LDS: a workgroup's slice of the table (128 KiB) + f16(scale x) of every column (2 G bytes).  T tokens
// need T x 2 G bytes of activations, so the columns are taken in PHASES: the layout orders every (slice, row) list by column
// window (VPTQ_SLICED_WINDOWS = 4 equal column ranges, `wstart` = where each begins inside the list), a phase stages the T
// tokens of 4 / phases windows - [column][token] halves, what one token of all columns takes - and every wave walks that
// part of its rows' lists.  Blocks of 64 elements do not end where windows do: a block that straddles two phases is walked in
// both, its elements outside the phase's columns read a zero column (so do the padding elements); with 4 phases a list of 16
// blocks costs about 3 more.  As many phases as the LDS asks for: 4096-column layers take 2 - 3 tokens in ONE.
//   * 2 tokens: per element one ds_read_b128 (entry), one ds_read_b32 (the column's two tokens), the entry converted to fp32
//     pairs once (+ the 256-entry residual entry: c + r in fp32) and one v_pk_fma_f32 per (token, pair of outputs); at the end of
//     a (row, phase) part the 64 lanes' sums are reduce-scattered (common.h:WaveReduce) into the row's sums in LDS.
//   * 3 - 4 tokens: the contraction over a block's 64 elements runs on the MATRIX PIPE - transposing gathers
//     (ds_read_b64_tr_b16) deliver entries and activations in the operand layout of v_mfma_f32_16x16x32, see `MF` below: a third
//     of the vector instructions, 4 sums per lane, no reduction (8192^2, 4 tokens: 35.5 -> 26.5 us; two tables 59.0 -> 40.4).
// The rows' sums live in LDS (owned by the wave: no atomics, a fixed order); after the last phase they go out as partial sums
// per (table, slice) and meet at the last workgroup of the row block, which adds them in a fixed order.  Folded arithmetic
// (gemv_k256m.hip): y[t] = sum c[idx] f16(s x[t]) + sum b x[t] + bias; EX (round 5): the reference's roundings per weight, see the
// kernel's comment.
// What the stream loop must NOT contain (found the hard way, profiles/r04/sliced_tokens_*.txt): a load the compiler can see
// (it then waits for vmcnt(0) in every step - the activations of a phase are loaded by ONE asm statement with its own wait,
// a layer's permutation is applied by a pre-pass), a load that is issued on some paths only (the waits then shrink down the
// unrolled round), 64-bit booleans and index -> address arithmetic in the scalar bookkeeping.
#include "sliced.h"

namespace vptq {

constexpr int kSTWindows = VPTQ_SLICED_WINDOWS;
constexpr int kSTRegRows = 4;   // rows per wave whose sums (4 registers each in matrix-pipe mode) are kept in registers
// element blocks in flight per wave (4 where a lane carries 64 sums: v = 16 with 4 tokens would spill at 8)
#ifndef VPTQ_ST_QUEUE
#define VPTQ_ST_QUEUE 8
#endif
template <int NV> constexpr int st_queue() { return NV >= 64 ? 4 : VPTQ_ST_QUEUE; }

// timing-only ablations (results wrong): bit 0 no FMAs, bit 1 no reduction at the end of a (row, phase) part, bit 2 no LDS gathers,
// bit 3 no staging of a later phase's activations (its barriers stay), bit 4 none of a later phase's barriers either
#ifndef VPTQ_ST_ABLATE
#define VPTQ_ST_ABLATE 0
#endif

typedef _Float16 st_h8_t __attribute__((ext_vector_type(8)));
typedef __bf16 st_b8_t __attribute__((ext_vector_type(8)));
template <typename DT>
static __device__ __forceinline__ f32x4 st_mfma(u32x4 a, u32x4 b, f32x4 c) {
  if constexpr (std::is_same<DT, F16>::value)
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(st_h8_t, a), __builtin_bit_cast(st_h8_t, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(st_b8_t, a), __builtin_bit_cast(st_b8_t, b), c, 0, 0, 0);
}
static __device__ __forceinline__ u32x2 st_lds_tr8(uint32_t byte_addr) {   // ds_read_b64_tr_b16
  typedef __attribute__((address_space(3))) s4_t lds_s4_t;
  return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_t*)(uintptr_t)byte_addr));
}

struct SlicedTokParams {
  SlicedParams p;            // as for one token; x / y: token 0, rows_per_wave / n_rowblocks: this launch's
  const int32_t* wstart;     // [slices][N][kSTWindows + 1]
  const int32_t* wstart2;    // second table
  const uint16_t* xs;        // the activations in COLUMN order: p.x, or the pre-pass's x[perm] (p.scale is in column order too)
  int tokens;                // 2 .. TOK
  int phases;                // 1, 2 or 4
  int wcols;                 // columns per layout window
  int x_stride, x_in_stride, y_stride;    // elements between two tokens of xs / p.x / y
  uint32_t bd_off, res_off, sum_off;   // LDS: sum b x parts [TOK][16 waves] floats; 256-entry residual table; row sums
  uint32_t sb_off;           // EX: LDS, per staged column 8 bytes {s, b, s, b} (and the zero column's zeros)
  int reg_sums;              // (matrix-pipe mode, <= kSTRegRows rows per wave) the rows' sums stay in registers: no LDS for them
};

// Up to kSTMaxGroup layers that read the SAME activations (q / k / v, gate / up) in one launch, as in gemv_sliced.hip: layer l
// owns the workgroups [start[l], start[l + 1]); the fixed part of a launch is paid once.
constexpr int kSTMaxGroup = 3;
struct SlicedTokGroupParams {
  int n;
  int start[kSTMaxGroup + 1];
  SlicedTokParams p[kSTMaxGroup];
};

// EX: the reference's roundings (vptq/ops/quant_gemm.py:121,155-156): every weight rebuilt as f16(f16(f16(c + r) s) + b) before it
// meets the RAW activations - matrix-pipe mode only (4 / 8 token slots), one table (+ the 256-entry residual table): the
// transposing gather that hands a lane component m of four elements' entries hands it the four elements' scales (m even) or biases
// (m odd) the same way - 8 bytes {s, b, s, b} per staged column; lanes m and m ^ 1 hold the same four elements, a swap inside
// the pair gives each both - so the rebuild is 2 packed ops per operand register (+ 1 for c + r, which takes the place of the
// second MFMA).  No sum b x part.
template <typename DT, int NSL, bool RES, int V, bool TWO, int TOK, bool EX = false>
__global__ __launch_bounds__(kSLThreads) void gemv_sliced_tok_kernel(const SlicedTokGroupParams GP) {
  // this workgroup's layer; its parameters come out of the kernel-argument segment through the scalar cache (gemv_sliced.hip)
  int layer = 0;
  if (GP.n > 1 && (int)blockIdx.x >= GP.start[1]) layer = 1;
  if (GP.n > 2 && (int)blockIdx.x >= GP.start[2]) layer = 2;
  layer = __builtin_amdgcn_readfirstlane(layer);
#if defined(__HIP_DEVICE_COMPILE__)
  typedef const char __attribute__((address_space(4)))* st_kernarg_t;
  typedef const SlicedTokParams __attribute__((address_space(4)))* st_params_t;
  const SlicedTokParams TP = *(st_params_t)((st_kernarg_t)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(SlicedTokGroupParams, p) +
                                            (size_t)layer * sizeof(SlicedTokParams));
#else
  const SlicedTokParams TP = GP.p[0];   // (host pass of the compiler: never executed)
#endif
  static_assert(((V == 8 && (NSL == 8 || NSL == 16)) || (V == 16 && (NSL == 16 || NSL == 32))) && (V == 8 || !RES) && !(TWO && RES) &&
                (TOK == 2 || TOK == 4 || TOK == 8) && (!EX || (TOK >= 4 && !TWO)), "instantiation");
  const SlicedParams& P = TP.p;
  constexpr int NSLT = TWO ? 2 * NSL : NSL;
  constexpr uint32_t kEntry = V * 2u;
  constexpr uint32_t kXStride = TOK * 2u;    // bytes per staged column
  constexpr int NV = TOK * V;                // sums per lane
  constexpr int kSTQueue = (TOK >= 4 && !(VPTQ_ST_ABLATE & 32)) ? 8 : st_queue<(NV > 64 ? 64 : NV)>();   // (MFMA mode: 4 sums per lane)
  const uint32_t kXOff = P.x_off;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  {
    typedef __attribute__((address_space(3))) unsigned char lds_u8_t;
    if ((uint32_t)(uintptr_t)(lds_u8_t*)smem != 0u) __builtin_trap();  // absolute LDS addressing
  }
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bx = (int)blockIdx.x - (layer == 0 ? 0 : layer == 1 ? GP.start[1] : GP.start[2]);
  const int sg = bx & (NSLT - 1), rb = bx / NSLT;
  const int s = sg & (NSL - 1);
  const bool second = TWO && sg >= NSL;
  const uint32_t* const elems_t = second ? P.elems2 : P.elems;
  const int32_t* const first_t = second ? P.first2 : P.first;
  const uint32_t* const cent_t = second ? P.cent2 : P.cent;
  const int32_t* const wstart_t = second ? TP.wstart2 : TP.wstart;
  const int N = P.N, G = P.G, tokens = TP.tokens;
  const int rpw = P.rows_per_wave;
  const int row0 = (rb * kSLWaves + wave) * rpw;
  const int n_rows = row0 >= N ? 0 : (N - row0 < rpw ? N - row0 : rpw);

  // ---- lane i: the list of row row0 + i in this slice - its first block and where the column windows begin
  int list_first = 0;
  int wofs[kSTWindows + 1];
#pragma unroll
  for (int j = 0; j <= kSTWindows; ++j) wofs[j] = 0;
  if (lane < n_rows) {
    const size_t li = (size_t)s * N + row0 + lane;
    list_first = as_global(first_t)[li];
#pragma unroll
    for (int j = 0; j <= kSTWindows; ++j) wofs[j] = as_global(wstart_t)[li * (kSTWindows + 1) + j];
  }

  // ---- this workgroup's part of its table into LDS by LDS-DMA (gemv_sliced.hip)
  {
    const uint32_t tab = second ? P.tab1 : P.tab0;
    const uint64_t va = (uint64_t)(uintptr_t)as_global(cent_t) + (uint64_t)s * (second ? P.stride1 : P.stride0) + (uint64_t)lane * 16u;
    for (uint32_t off = (uint32_t)wave * 1024u; off < tab; off += kSLWaves * 1024u) {
      const uint32_t d = (uint32_t)__builtin_amdgcn_readfirstlane((int)off);
      if (off + (uint32_t)lane * 16u < tab) {
        const uint64_t v = va + (uint64_t)off;
        uint32_t keep_m0;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep_m0) : "v"(v), "s"(d) : "memory");
      }
    }
  }
  if constexpr (RES) {
    if (wave < 4) {
      const uint64_t v = (uint64_t)(uintptr_t)as_global(P.rcent) + (uint64_t)wave * 1024u + (uint64_t)lane * 16u;
      const uint32_t d = (uint32_t)__builtin_amdgcn_readfirstlane((int)(TP.res_off + (uint32_t)wave * 1024u));
      uint32_t keep_m0;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep_m0) : "v"(v), "s"(d) : "memory");
    }
  }
  // ---- sum b x per token (the workgroups of slice 0: it rides in their partial sums), x in input-feature order
  if (sg == 0 && !EX) {
    float bd[TOK];
#pragma unroll
    for (int t = 0; t < TOK; ++t) bd[t] = 0.f;
    if (P.wbias != nullptr) {
      for (int q = tid; q < (G >> 3); q += kSLThreads) {
        const u32x4 bv = *(const u32x4*)(as_global(P.wbias) + 8 * q);
#pragma unroll
        for (int t = 0; t < TOK; ++t) {
          if (t < tokens) {
            const u32x4 xv = *(const u32x4*)(as_global(P.x) + (size_t)t * TP.x_in_stride + 8 * q);
#pragma unroll
            for (int i = 0; i < 4; ++i) bd[t] = DT::dot2(xv[i], bv[i], bd[t]);
          }
        }
      }
    }
#pragma unroll
    for (int t = 0; t < TOK; ++t) {
      const float v = wave_sum(bd[t]);
      if (lane == 0) *(float*)(smem + TP.bd_off + (uint32_t)(t * kSLWaves + wave) * 4u) = v;
    }
  }
  // ---- the rows' sums start at zero
  {
    typedef __attribute__((address_space(3))) u32x4 lds_q_t;
    const int n16 = TP.reg_sums ? 0 : kSLWaves * rpw * NV / 4;
    for (int q = tid; q < n16; q += kSLThreads) *(lds_q_t*)(uintptr_t)(TP.sum_off + (uint32_t)q * 16u) = u32x4{0u, 0u, 0u, 0u};
  }

  // ---- a phase's columns [c0, c0 + wlen): its windows of every list
  const int wpp = kSTWindows / TP.phases;   // layout windows per phase
  auto phase_c0 = [&](int ph) { const int c = ph * wpp * TP.wcols; return c < G ? c : G; };
  auto phase_c1 = [&](int ph) { const int c = (ph + 1) * wpp * TP.wcols; return (ph == TP.phases - 1 || c > G) ? G : c; };
  // the 8 columns q of a phase for every token, scaled.  The loads AND their wait are one asm statement: a load the compiler
  // could see inside the stream loop makes it lose count of the queue's loads (every step then waits for vmcnt(0): the first
  // versions ran with NO load in flight across a step - 25 of 37 us for 4 tokens at 8192^2).  The statement's wait drains the
  // queue once per phase; the steps keep their counted waits.  (x is in column order here: a layer's permutation is applied
  // by a pre-pass, launch_gemv_sliced_tok.)
  struct Chunk { u32x4 v[TOK]; u32x4 s, b; };
  auto load_chunk = [&](int c0, int q) __attribute__((always_inline)) {
    Chunk ch;
    const int c = c0 + 8 * q;
    const uint16_t* const ps = as_global(P.scale) + c;
    const uint16_t* px[TOK];
#pragma unroll
    for (int t = 0; t < TOK; ++t) px[t] = as_global(TP.xs) + (size_t)(t < tokens ? t : 0) * TP.x_stride + c;
    u32x4 sv;
    // EX: the column-order bias rides in the same statement (ONE exposed wait per chunk; without a bias: the scales again, dropped)
    [[maybe_unused]] u32x4 bv = {0u, 0u, 0u, 0u};
    [[maybe_unused]] const uint16_t* const pb = (EX && P.cbias != nullptr) ? as_global(P.cbias) + c : ps;
    if constexpr (TOK == 2) {
      u32x4 a, b;
      asm volatile("global_load_dwordx4 %0, %3, off\n\tglobal_load_dwordx4 %1, %4, off\n\tglobal_load_dwordx4 %2, %5, off\n\ts_waitcnt vmcnt(0)"
                   : "=&v"(a), "=&v"(b), "=&v"(sv) : "v"(px[0]), "v"(px[1]), "v"(ps) : "memory");
      ch.v[0] = a; ch.v[1] = b;
    } else if constexpr (TOK == 4) {
      u32x4 a, b, c2, d;
      if constexpr (EX)
        asm volatile("global_load_dwordx4 %0, %6, off\n\tglobal_load_dwordx4 %1, %7, off\n\tglobal_load_dwordx4 %2, %8, off\n\t"
                     "global_load_dwordx4 %3, %9, off\n\tglobal_load_dwordx4 %4, %10, off\n\tglobal_load_dwordx4 %5, %11, off\n\t"
                     "s_waitcnt vmcnt(0)"
                     : "=&v"(a), "=&v"(b), "=&v"(c2), "=&v"(d), "=&v"(sv), "=&v"(bv)
                     : "v"(px[0]), "v"(px[1]), "v"(px[2]), "v"(px[3]), "v"(ps), "v"(pb) : "memory");
      else
        asm volatile("global_load_dwordx4 %0, %5, off\n\tglobal_load_dwordx4 %1, %6, off\n\tglobal_load_dwordx4 %2, %7, off\n\t"
                     "global_load_dwordx4 %3, %8, off\n\tglobal_load_dwordx4 %4, %9, off\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(a), "=&v"(b), "=&v"(c2), "=&v"(d), "=&v"(sv)
                     : "v"(px[0]), "v"(px[1]), "v"(px[2]), "v"(px[3]), "v"(ps) : "memory");
      ch.v[0] = a; ch.v[1] = b; ch.v[2] = c2; ch.v[3] = d;
    } else {   // 8 token slots: two statements (ONE with ten loads - 40 result registers live across the wait - was measured at
      // 95 instead of 55 us per 8192^2 layer: profiles/r05/sliced_tokens_exact.txt); the bias rides in the second
      u32x4 a, b, c2, d, e2, f2, g2, h2;
      asm volatile("global_load_dwordx4 %0, %5, off\n\tglobal_load_dwordx4 %1, %6, off\n\tglobal_load_dwordx4 %2, %7, off\n\t"
                   "global_load_dwordx4 %3, %8, off\n\tglobal_load_dwordx4 %4, %9, off\n\ts_waitcnt vmcnt(0)"
                   : "=&v"(a), "=&v"(b), "=&v"(c2), "=&v"(d), "=&v"(sv)
                   : "v"(px[0]), "v"(px[1]), "v"(px[2]), "v"(px[3]), "v"(ps) : "memory");
      ch.v[0] = a; ch.v[1] = b; ch.v[2] = c2; ch.v[3] = d;
      if constexpr (EX)
        asm volatile("global_load_dwordx4 %0, %5, off\n\tglobal_load_dwordx4 %1, %6, off\n\tglobal_load_dwordx4 %2, %7, off\n\t"
                     "global_load_dwordx4 %3, %8, off\n\tglobal_load_dwordx4 %4, %9, off\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(e2), "=&v"(f2), "=&v"(g2), "=&v"(h2), "=&v"(bv)
                     : "v"(px[TOK - 4]), "v"(px[TOK - 3]), "v"(px[TOK - 2]), "v"(px[TOK - 1]), "v"(pb) : "memory");
      else
        asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %5, off\n\tglobal_load_dwordx4 %2, %6, off\n\t"
                     "global_load_dwordx4 %3, %7, off\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(e2), "=&v"(f2), "=&v"(g2), "=&v"(h2)
                     : "v"(px[TOK - 4]), "v"(px[TOK - 3]), "v"(px[TOK - 2]), "v"(px[TOK - 1]) : "memory");
      ch.v[TOK - 4] = e2; ch.v[TOK - 3] = f2; ch.v[TOK - 2] = g2; ch.v[TOK - 1] = h2;
    }
    if constexpr (EX) {   // the activations stay raw; scale and bias (column order) are staged per column
      if (P.cbias == nullptr) bv = u32x4{0u, 0u, 0u, 0u};
      ch.s = sv; ch.b = bv;
#pragma unroll
      for (int t = 0; t < TOK; ++t) {
#pragma unroll
        for (int i = 0; i < 4; ++i) ch.v[t][i] = t < tokens ? ch.v[t][i] : 0u;
      }
      return ch;
    }
#pragma unroll
    for (int t = 0; t < TOK; ++t) {
#pragma unroll
      for (int i = 0; i < 4; ++i) ch.v[t][i] = t < tokens ? DT::mul2(ch.v[t][i], sv[i]) : 0u;
    }
    return ch;
  };
  // [column][token] halves: column 2 i of the chunk = the low halves of v[t][i], column 2 i + 1 the high halves
  auto store_chunk = [&](const Chunk& ch, int q) __attribute__((always_inline)) {
    const uint32_t base = kXOff + (uint32_t)q * 8u * kXStride;
    if constexpr (EX) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const uint32_t sw = (j & 1) ? (ch.s[j / 2] >> 16) : (ch.s[j / 2] & 0xffffu);
        const uint32_t bw = (j & 1) ? (ch.b[j / 2] >> 16) : (ch.b[j / 2] & 0xffffu);
        const uint32_t sb = sw | (bw << 16);
        *(__attribute__((address_space(3))) u32x2*)(uintptr_t)(TP.sb_off + (uint32_t)(q * 8 + j) * 8u) = u32x2{sb, sb};
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if constexpr (TOK == 2) {
        const uint32_t lo = (ch.v[0][i] & 0xffffu) | (ch.v[1][i] << 16);
        const uint32_t hi = (ch.v[0][i] >> 16) | (ch.v[1][i] & 0xffff0000u);
        *(__attribute__((address_space(3))) u32x2*)(uintptr_t)(base + (uint32_t)i * 8u) = u32x2{lo, hi};
      } else if constexpr (TOK == 4) {
        const uint32_t lo01 = (ch.v[0][i] & 0xffffu) | (ch.v[1][i] << 16), lo23 = (ch.v[2][i] & 0xffffu) | (ch.v[3][i] << 16);
        const uint32_t hi01 = (ch.v[0][i] >> 16) | (ch.v[1][i] & 0xffff0000u), hi23 = (ch.v[2][i] >> 16) | (ch.v[3][i] & 0xffff0000u);
        lds_store16(base + (uint32_t)i * 16u, u32x4{lo01, lo23, hi01, hi23});
      } else {   // 8 tokens: 16 bytes per column
        u32x4 lo, hi;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          lo[p] = (ch.v[(2 * p) % TOK][i] & 0xffffu) | (ch.v[(2 * p + 1) % TOK][i] << 16);
          hi[p] = (ch.v[(2 * p) % TOK][i] >> 16) | (ch.v[(2 * p + 1) % TOK][i] & 0xffff0000u);
        }
        lds_store16(base + (uint32_t)(2 * i) * 16u, lo);
        lds_store16(base + (uint32_t)(2 * i + 1) * 16u, hi);
      }
    }
  };

  // 8 token slots: a chunk is staged in two HALVES of four tokens (8 of a column's 16 bytes each).  All eight tokens (+ scale
  // + bias: 40 result registers and 10 addresses across one wait, next to the queue's registers) spilled into the stream loop:
  // the 256-entry-table variants ran at 90 instead of 55 us per 8192^2 layer (profiles/r05/sliced_tokens_exact.txt)
  [[maybe_unused]] auto stage8 = [&](int c0, int q) __attribute__((always_inline)) {
    const int c = c0 + 8 * q;
    const uint16_t* const ps = as_global(P.scale) + c;
    const uint16_t* const pb = (EX && P.cbias != nullptr) ? as_global(P.cbias) + c : ps;
    const uint32_t base = kXOff + (uint32_t)q * 8u * kXStride;
    u32x4 sv = {0u, 0u, 0u, 0u}, bv = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const uint16_t* px[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) px[t] = as_global(TP.xs) + (size_t)(4 * hf + t < tokens ? 4 * hf + t : 0) * TP.x_stride + c;
      u32x4 v[4], w;
      asm volatile("global_load_dwordx4 %0, %5, off\n\tglobal_load_dwordx4 %1, %6, off\n\tglobal_load_dwordx4 %2, %7, off\n\t"
                   "global_load_dwordx4 %3, %8, off\n\tglobal_load_dwordx4 %4, %9, off\n\ts_waitcnt vmcnt(0)"
                   : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(w)
                   : "v"(px[0]), "v"(px[1]), "v"(px[2]), "v"(px[3]), "v"(hf == 0 ? ps : pb) : "memory");
      if (hf == 0) sv = w; else bv = w;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[t][i] = 4 * hf + t < tokens ? (EX ? v[t][i] : DT::mul2(v[t][i], sv[i])) : 0u;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {   // columns 2 i (low halves) and 2 i + 1 (high halves): tokens 4 hf .. 4 hf + 3 = bytes 8 hf .. of the 16
        const uint32_t lo01 = (v[0][i] & 0xffffu) | (v[1][i] << 16), lo23 = (v[2][i] & 0xffffu) | (v[3][i] << 16);
        const uint32_t hi01 = (v[0][i] >> 16) | (v[1][i] & 0xffff0000u), hi23 = (v[2][i] >> 16) | (v[3][i] & 0xffff0000u);
        *(__attribute__((address_space(3))) u32x2*)(uintptr_t)(base + (uint32_t)(2 * i) * 16u + 8u * (uint32_t)hf) = u32x2{lo01, lo23};
        *(__attribute__((address_space(3))) u32x2*)(uintptr_t)(base + (uint32_t)(2 * i + 1) * 16u + 8u * (uint32_t)hf) = u32x2{hi01, hi23};
      }
    }
    if constexpr (EX) {
      if (P.cbias == nullptr) bv = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const uint32_t sw = (j & 1) ? (sv[j / 2] >> 16) : (sv[j / 2] & 0xffffu);
        const uint32_t bw = (j & 1) ? (bv[j / 2] >> 16) : (bv[j / 2] & 0xffffu);
        const uint32_t sb = sw | (bw << 16);
        *(__attribute__((address_space(3))) u32x2*)(uintptr_t)(TP.sb_off + (uint32_t)(q * 8 + j) * 8u) = u32x2{sb, sb};
      }
    }
  };

  // ---- element queue.  The stream of a wave = phase by phase, row by row, the phase's part of the row's list; the ISSUE side
  // walks it ahead of the CONSUME side by kSTQueue blocks and does not stop at a phase's end (a first version started every
  // phase's stream behind its barriers: one exposed memory latency per phase, 2.5 - 3 us each)
  // MF (v = 8, 4 token slots): the contraction over a block's 64 elements runs on the matrix pipe.  ds_read_b64_tr_b16 hands a
  // lane component m of FOUR elements' entries (gemm_k256t.hip: inside 16 lanes, source lane 4 e + c supplies an 8-byte chunk's
  // address, result lane 4 c' + m receives half m of the chunks of source lanes 4 e' + c', e' = 0..3) - the operand layout of
  // v_mfma_f32_16x16x32 (lane 16 g + j: row / column j, K = 8 g .. 8 g + 7).  Source lane (g, e, c) of read i = 0, 1 takes
  // the element at position 32 i + 8 g + 2 e + (c >> 1) of the block and chunk c & 1 of its entry, so a result lane's column
  // is 8 h + component, h = c' >> 1 the element SET; the activations go through the same read - chunk = the column's 4 tokens
  // (c & 1 = 0) or the zero column - so a row is 8 h + token.  D[8 h + token][8 h' + component] is the sum over the block's
  // K = (g, i, e) for h = h' (the two diagonal blocks; the others mix sets and are dropped): lanes 0 - 7 hold set 0, lanes
  // 40 - 47 set 1, 4 registers = 4 tokens.  Per block 4 (6 with the 256-entry residual table) gathers + 1 (2) MFMAs instead of
  // 2 gathers + 12 conversions + 16 packed FMAs, and the end of a (row, phase) part is a lane swap and an add instead of a
  // 64-lane reduce-scatter of 32 sums.  Every lane loads TWO element words per block (a pair of lanes the same ones).
  // v = 16: an entry is FOUR chunks, so the 16 columns are the 16 components of ONE set, source lane (g, e, c) of read i = 0..3
  // takes the element at position 16 i + 4 g + e and chunk c of its entry (activations: c = 0 the tokens, else the zero column),
  // reads 0, 1 feed one MFMA, reads 2, 3 a second one; lanes 0 - 15 hold the sums, four element words per lane and block.
  constexpr bool MF = (TOK >= 4 && !(VPTQ_ST_ABLATE & 32)) || TOK == 8;   // (8 token slots exist on the matrix pipe only)
  constexpr int EW = MF ? (V == 8 ? 2 : 4) : 1;
  constexpr int EPR = 64 / EW;                   // elements per read of the block
  uint32_t eq[kSTQueue][EW];
  uint32_t rq[RES ? kSTQueue : 1][EW];
  const int epos = MF ? (V == 8 ? (lane >> 1) : (lane >> 2)) : lane;      // this lane's (first) element of a block
  // lane i: the blocks [first, first + cnt) of row row0 + i that hold the phase's windows (a block that holds a window's edge
  // is walked in both phases)
  auto seg_of = [&](int ph, int& first, int& cnt) __attribute__((always_inline)) {
    int ws = wofs[0], we = wofs[kSTWindows];
#pragma unroll
    for (int j = 0; j <= kSTWindows; ++j) {
      if (j == ph * wpp) ws = wofs[j];
      if (j == (ph + 1) * wpp) we = wofs[j];
    }
    cnt = we > ws ? ((we + 63) >> 6) - (ws >> 6) : 0;
    first = list_first + (ws >> 6);
  };
  // The bookkeeping of a step is SCALAR work, and a SIMD issues one scalar instruction per 4 cycles for its 4 waves just as it
  // issues one vector instruction: ~30 scalar instructions per step (64-bit booleans, block index -> address, two flags per side)
  // were as much of the walk as the vector work.  Now: a running scalar pointer per side, one count-down, one stride that
  // turns 0 at the end of the stream.
  int iseg_first = 0, iseg_cnt = 0, cseg_cnt = 0, unused_first = 0;
  int iq_ph = 0, iq_row = 0, iq_left = 0;        // issue side: phase, row inside the wave, blocks left in its part
  int cq_ph = 0, cq_row = 0, cq_left = 0;        // consume side
  int done = 0;
  const uint32_t* iq_ptr = as_global(elems_t);   // the next block (wave-uniform)
  const uint8_t* iq_rptr = RES ? as_global(P.res) : nullptr;
  int iq_stride = 64;                            // elements to the next block: 0 once the stream has ended (its last block again)
  seg_of(0, iseg_first, iseg_cnt);
  cseg_cnt = iseg_cnt;
  // the next (phase, row) that has blocks
  auto iq_advance = [&]() __attribute__((always_inline)) {
    for (;;) {
      while (iq_row < n_rows) {
        const int c = __builtin_amdgcn_readlane(iseg_cnt, iq_row);
        if (c != 0) {
          iq_left = c;
          const size_t b = (size_t)__builtin_amdgcn_readlane(iseg_first, iq_row) * 64;
          iq_ptr = as_global(elems_t) + b;
          if constexpr (RES) iq_rptr = as_global(P.res) + b;
          return;
        }
        ++iq_row;
      }
      if (iq_ph + 1 >= TP.phases) { iq_stride = 0; iq_left = 0x7fffffff; return; }   // (the pointers stay on the last block)
      ++iq_ph;
      seg_of(iq_ph, iseg_first, iseg_cnt);
      iq_row = 0;
    }
  };
  auto issue = [&](auto slot_c) __attribute__((always_inline)) {
    constexpr int S = decltype(slot_c)::value;
#pragma unroll
    for (int i = 0; i < EW; ++i) {
      eq[S][i] = __builtin_nontemporal_load(iq_ptr + epos + EPR * i);
      if constexpr (RES) rq[S][i] = iq_rptr[epos + EPR * i];
    }
    if (--iq_left != 0) {
      iq_ptr += iq_stride;
      if constexpr (RES) iq_rptr += iq_stride;
    } else {
      ++iq_row;
      iq_advance();
    }
  };
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  f32x2_t acc2[TOK][V / 2];
#pragma unroll
  for (int t = 0; t < TOK; ++t)
#pragma unroll
    for (int i = 0; i < V / 2; ++i) acc2[t][i] = f32x2_t{0.f, 0.f};
  f32x4 accm = {0.f, 0.f, 0.f, 0.f};   // MF: D of the MFMA (rows = 4 (lane / 16) + r, column = lane % 16)
  f32x4 saved[MF ? kSTRegRows : 1];    // MF, reg_sums: the sums of the wave's rows over the phases
#pragma unroll
  for (int k = 0; k < (MF ? kSTRegRows : 1); ++k) saved[k] = f32x4{0.f, 0.f, 0.f, 0.f};
  uint32_t c0 = 0, wlen = 0;   // the consume side's phase: its columns
  // a (row, phase) part ends: the 64 lanes' sums -> the row's sums in LDS (this wave owns them)
  auto row_end = [&]() __attribute__((always_inline)) {
    if constexpr (MF) {
      if (TP.reg_sums) {   // D as it is (v = 8: both sets' lanes) into the row's registers; the sets meet once, at the end
#pragma unroll
        for (int k = 0; k < kSTRegRows; ++k)
          if (cq_row == k) saved[k] += accm;
        accm = f32x4{0.f, 0.f, 0.f, 0.f};
        return;
      }
      // v = 8: lanes 0 - 7: set 0's sums of component `lane` for tokens r = 0..3; lanes 40 - 47: set 1's.  Lane 8 + j gets lane
      // 40 + j's by the 32-lane swap, lane j lane 8 + j's by a rotation inside the row of 16.  v = 16: lanes 0 - 15 have them.
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = accm[r];
        if constexpr (V == 8) {
          auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(accm[r]), __float_as_uint(accm[r]), false, false);
          const float hi = __uint_as_float(sw[1]);   // (lanes 0 - 31: the value of lane + 32)
          v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(hi), 0x128 /* row_ror:8 */, 0xf, 0xf, false));
        }
        // (8 token slots: lanes 16 .. 16 + v - 1 hold tokens 4 - 7 of the same outputs)
        if ((lane & 15) < V && lane < (TOK == 8 ? 32 : 16)) {
          typedef __attribute__((address_space(3))) float lds_f_t;
          lds_f_t* const sp = (lds_f_t*)(uintptr_t)(TP.sum_off + (uint32_t)(((wave * rpw + cq_row) * NV) + (4 * (lane >> 4) + r) * V + (lane & 15)) * 4u);
          *sp = *sp + v;
        }
      }
      accm = f32x4{0.f, 0.f, 0.f, 0.f};
      return;
    } else {
    float acc[NV];
#pragma unroll
    for (int t = 0; t < TOK; ++t)
#pragma unroll
      for (int i = 0; i < V / 2; ++i) { acc[t * V + 2 * i] = acc2[t][i][0]; acc[t * V + 2 * i + 1] = acc2[t][i][1]; }
    if constexpr ((VPTQ_ST_ABLATE & 2) == 0) WaveReduce<NV>::run(acc, lane);
    constexpr int kShift = WaveReduce<NV>::kShift;
    if ((lane & ((1 << kShift) - 1)) == 0) {
      typedef __attribute__((address_space(3))) float lds_f_t;
      lds_f_t* const sp = (lds_f_t*)(uintptr_t)(TP.sum_off + (uint32_t)(((wave * rpw + cq_row) * NV) + (lane >> kShift)) * 4u);
      *sp = *sp + acc[0];
    }
#pragma unroll
    for (int t = 0; t < TOK; ++t)
#pragma unroll
      for (int i = 0; i < V / 2; ++i) acc2[t][i] = f32x2_t{0.f, 0.f};
    }
  };
  auto consume = [&](auto slot_c) __attribute__((always_inline)) {
    constexpr int S = decltype(slot_c)::value;
    if constexpr (MF) {
      if constexpr ((VPTQ_ST_ABLATE & 4) != 0) { accm[0] += __uint_as_float(eq[S][0] ^ eq[S][EW - 1]); return; }
      constexpr uint32_t kChunks = V / 4;                                    // 8-byte chunks of an entry
      const uint32_t q8 = (uint32_t)(lane & (kChunks - 1)) * 8u;             // this lane's chunk
      // chunk 0's lanes bring tokens 0 - 3 (with 8 token slots chunk 1's lanes tokens 4 - 7), the others the zero column
      const uint32_t xq = TOK == 8 ? (uint32_t)(lane & 1) * 8u : 0u;
      const uint32_t qmask = (lane & (kChunks - 1)) >= (TOK == 8 ? 2u : 1u) ? 0xffffffffu : 0u;
      u32x2 at[EW], bt[EW], rt[RES ? EW : 1];
      [[maybe_unused]] u32x2 sc[EX ? EW : 1], bi[EX ? EW : 1];
#pragma unroll
      for (int i = 0; i < EW; ++i) {
        const uint32_t e = eq[S][i];
        bt[i] = st_lds_tr8((e >> 16) * kEntry + q8);
        const uint32_t cc = (e & 0xffffu) - c0;
        uint32_t ci = cc | qmask;
        ci = ci < wlen ? ci : wlen;
        at[i] = st_lds_tr8(kXOff + ci * kXStride + xq);
        if constexpr (RES) rt[i] = st_lds_tr8(TP.res_off + (rq[S][i] << 4) + q8);
        if constexpr (EX) {   // EVERY chunk's lanes bring their element's scale and bias (the zero column's: 0, 0)
          // lanes with m = lane & 3 even receive the four elements' scales, odd ones their biases; the neighbour has the other
          const u32x2 t = st_lds_tr8(TP.sb_off + (cc < wlen ? cc : wlen) * 8u);
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)t[h], 0xB1 /* quad_perm [1, 0, 3, 2] */, 0xf, 0xf, true);
            sc[i][h] = (lane & 1) ? o : t[h];
            bi[i][h] = (lane & 1) ? t[h] : o;
          }
        }
      }
      if constexpr (EX) {
#pragma unroll
        for (int i = 0; i < EW; ++i) {
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            uint32_t w = bt[i][h];
            if constexpr (RES) w = DT::add2(w, rt[i][h]);
            bt[i][h] = DT::add2(DT::mul2(w, sc[i][h]), bi[i][h]);
          }
        }
      }
#pragma unroll
      for (int i = 0; i < EW; i += 2) {
        const u32x4 A = {at[i][0], at[i][1], at[i + 1][0], at[i + 1][1]};
        accm = st_mfma<DT>(A, u32x4{bt[i][0], bt[i][1], bt[i + 1][0], bt[i + 1][1]}, accm);
        if constexpr (RES && !EX) accm = st_mfma<DT>(A, u32x4{rt[i][0], rt[i][1], rt[i + 1][0], rt[i + 1][1]}, accm);
      }
      return;
    }
    const uint32_t e = eq[S][0];
    if constexpr ((VPTQ_ST_ABLATE & 4) != 0) { acc2[0][0][0] += __uint_as_float(e); return; }
    constexpr int W4 = V / 8;
    u32x4 ent[W4];
    const uint32_t ea = (e >> 16) * kEntry;
#pragma unroll
    for (int w = 0; w < W4; ++w) ent[w] = lds_load16(ea + 16u * (uint32_t)w);
    // the element's column inside the phase, or the zero column behind it (other phases' elements, padding)
    uint32_t ci = (e & 0xffffu) - c0;
    ci = ci < wlen ? ci : wlen;
    uint32_t xw[TOK / 2];
    if constexpr (TOK == 2) {
      xw[0] = *(const __attribute__((address_space(3))) uint32_t*)(uintptr_t)(kXOff + ci * kXStride);
    } else {
      const u32x2 t = *(const __attribute__((address_space(3))) u32x2*)(uintptr_t)(kXOff + ci * kXStride);
      xw[0] = t[0]; xw[1] = t[1];
    }
    u32x4 rent = {0u, 0u, 0u, 0u};
    if constexpr (RES) rent = lds_load16(TP.res_off + (rq[S][0] << 4));
    if constexpr ((VPTQ_ST_ABLATE & 1) != 0) {
      acc2[0][0][0] += __uint_as_float(ent[0][0] ^ xw[0] ^ rent[0]);
      return;
    }
    // the entry (+ its residual entry: c + r once, in fp32) as fp32 pairs, the tokens' activations as fp32 pairs, then one
    // packed FMA per (token, pair of outputs): v_pk_fma_f32 takes the activation from either half of its pair (op_sel)
    f32x2_t ef[V / 2], xf[TOK / 2];
#pragma unroll
    for (int i = 0; i < V / 2; ++i) {
      const uint32_t ew = ent[i / 4][i % 4];
      if constexpr (std::is_same<DT, F16>::value) {
        if constexpr (RES) {
          const uint32_t rw = rent[i % 4];
          float lo, hi;
          asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(lo) : "v"(ew), "v"(rw));
          asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(hi) : "v"(ew), "v"(rw));
          ef[i] = f32x2_t{lo, hi};
        } else {
          ef[i] = f32x2_t{DT::lo(ew), DT::hi(ew)};
        }
      } else {
        ef[i] = f32x2_t{DT::to_float((uint16_t)(ew & 0xffffu)), DT::to_float((uint16_t)(ew >> 16))};
        if constexpr (RES) {
          const uint32_t rw = rent[i % 4];
          ef[i] += f32x2_t{DT::to_float((uint16_t)(rw & 0xffffu)), DT::to_float((uint16_t)(rw >> 16))};
        }
      }
    }
#pragma unroll
    for (int t = 0; t < TOK / 2; ++t)
      xf[t] = f32x2_t{DT::to_float((uint16_t)(xw[t] & 0xffffu)), DT::to_float((uint16_t)(xw[t] >> 16))};
#pragma unroll
    for (int t = 0; t < TOK; ++t) {
#pragma unroll
      for (int i = 0; i < V / 2; ++i) {
        f32x2_t a = acc2[t][i];   // (an asm operand cannot name a captured array element)
        const f32x2_t ev = ef[i], xv = xf[t / 2];
        if (t % 2 == 0) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "+v"(a) : "v"(ev), "v"(xv));
        else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(a) : "v"(ev), "v"(xv));
        acc2[t][i] = a;
      }
    }
  };
  // the consume side enters phase ph: everybody has left the previous phase's activations (barrier), this thread's chunks of
  // the new ones go to LDS, barrier
  auto enter_phase = [&](int ph) __attribute__((always_inline)) {
    if ((VPTQ_ST_ABLATE & 8) && ph > 0) {
      c0 = (uint32_t)phase_c0(ph);
      wlen = (uint32_t)phase_c1(ph) - c0;
      if (!(VPTQ_ST_ABLATE & 16)) {
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_barrier();
      }
      return;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    c0 = (uint32_t)phase_c0(ph);
    wlen = (uint32_t)phase_c1(ph) - c0;
    const int chunks = (int)(wlen >> 3);
    for (int q = tid; q < chunks; q += kSLThreads) {
      if constexpr (TOK == 8) stage8((int)c0, q);
      else store_chunk(load_chunk((int)c0, q), q);
    }
    if (tid == 0) {   // the zero column
      if constexpr (TOK == 2) *(__attribute__((address_space(3))) uint32_t*)(uintptr_t)(kXOff + wlen * kXStride) = 0u;
      else if constexpr (TOK == 4) *(__attribute__((address_space(3))) u32x2*)(uintptr_t)(kXOff + wlen * kXStride) = u32x2{0u, 0u};
      else lds_store16(kXOff + wlen * kXStride, u32x4{0u, 0u, 0u, 0u});
      if constexpr (EX) *(__attribute__((address_space(3))) u32x2*)(uintptr_t)(TP.sb_off + wlen * 8u) = u32x2{0u, 0u};   // (scale 0, bias 0: a weight of 0)
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
  };
  // the consume side's next row with blocks; at the end of a phase's rows the next phase is entered (EVERY wave enters every
  // phase: the barriers), at the end of the last one the wave is done
  auto cq_advance = [&]() __attribute__((always_inline)) {
    for (;;) {
      while (cq_row < n_rows) {
        cq_left = __builtin_amdgcn_readlane(cseg_cnt, cq_row);
        if (cq_left != 0) return;
        ++cq_row;
      }
      if (cq_ph + 1 >= TP.phases) { done = 1; cq_left = 0x7fffffff; return; }
      ++cq_ph;
      seg_of(cq_ph, unused_first, cseg_cnt);
      cq_row = 0;
      enter_phase(cq_ph);
    }
  };
  auto step = [&](auto slot_c) __attribute__((always_inline)) {
    __builtin_amdgcn_sched_barrier(0);
    // (after the wave's last block the steps of the round still issue their load - the last block again -: ONE load per
    // step on every path is what lets the compiler count: vmcnt(queue - 1) before every consume.  With the load under
    // `if (!done)` it waited for vmcnt(7), 6, ... 0 down the round: half the queue on average, drained once per round.)
    if (done == 0) consume(slot_c);
    __builtin_amdgcn_sched_barrier(0);
    issue(slot_c);
    __builtin_amdgcn_sched_barrier(0);
    if (--cq_left == 0) {   // (a wave that is done counts down from 2^31)
      row_end();
      ++cq_row;
      cq_advance();
    }
  };

  iq_advance();
  sl_for_slots<kSTQueue>([&](auto slot_c) { issue(slot_c); __builtin_amdgcn_sched_barrier(0); });
  asm volatile("s_waitcnt vmcnt(%0)" :: "n"(kSTQueue * (RES ? 2 : 1) * EW) : "memory");   // the DMA'd table: older than the queue's loads
  enter_phase(0);
  cq_advance();
  while (done == 0) sl_for_slots<kSTQueue>(step);

  // ---- partial sums of this (table, slice): [token][NSLT][N x V]; sum b x rides with slice 0
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
  const int rows_wg = kSLWaves * rpw;
  const int r_first = rb * rows_wg;
  const int n_rows_wg = N - r_first < rows_wg ? N - r_first : rows_wg;
  {
    float bdot[TOK];
#pragma unroll
    for (int t = 0; t < TOK; ++t) {
      bdot[t] = 0.f;
      if (sg == 0 && !EX) {
        const float* const bp = (const float*)(smem + TP.bd_off) + t * kSLWaves;
#pragma unroll
        for (int i = 0; i < kSLWaves; ++i) bdot[t] += bp[i];
      }
    }
    if constexpr (MF) {
      if (TP.reg_sums) {   // every wave stores its rows' sums itself: lanes 0 .. v - 1 = the outputs, 4 registers = the tokens
#pragma unroll
        for (int k = 0; k < kSTRegRows; ++k) {
          if (k < n_rows) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float v = saved[k][r];
              if constexpr (V == 8) {
                auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
                v += __int_as_float(__builtin_amdgcn_update_dpp(0, (int)sw[1], 0x128 /* row_ror:8 */, 0xf, 0xf, false));
              }
              const int tk = 4 * (lane >> 4) + r;     // (8 token slots: lanes 16 .. hold tokens 4 - 7)
              if ((lane & 15) < V && lane < (TOK == 8 ? 32 : 16) && tk < tokens) {
                float bt = bdot[r];
                if constexpr (TOK == 8) bt = (lane >> 4) ? bdot[(4 + r) % TOK] : bdot[r];
                float* const pp = as_global(P.partial) + (((size_t)tk * NSLT + sg) * N + (size_t)(row0 + k)) * V + (lane & 15);
                __hip_atomic_store(pp, v + bt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              }
            }
          }
        }
      }
    }
    for (int k = tid; k < (TP.reg_sums ? 0 : n_rows_wg * NV); k += kSLThreads) {
      const int r = k / NV, e = k % NV, t = e / V, o = e % V;
      if (t < tokens) {
        float v = *(const float*)(smem + TP.sum_off + (uint32_t)k * 4u);
#pragma unroll
        for (int tt = 0; tt < TOK; ++tt) v += tt == t ? bdot[tt] : 0.f;
        float* const pp = as_global(P.partial) + (((size_t)t * NSLT + sg) * N + (size_t)(r_first + r)) * V + o;
        __hip_atomic_store(pp, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // write-through (sc1): read on another XCD
      }
    }
  }
  // ---- the last workgroup of the row block adds the slices (gemv_sliced.hip)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  uint32_t* const flag = (uint32_t*)(smem + TP.bd_off);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  if (tid == 0) {
    const uint32_t before = __hip_atomic_fetch_add(as_global(P.arrived) + rb, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t lastone = before == (uint32_t)NSLT - 1u ? 1u : 0u;
    if (lastone) __hip_atomic_store(as_global(P.arrived) + rb, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *flag = lastone;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
  if (*flag == 0u) return;
  const int n_out = n_rows_wg * V;
  for (int k = tid; k < n_out * tokens; k += kSLThreads) {
    const int t = k / n_out;
    const size_t o = (size_t)r_first * V + (size_t)(k - t * n_out);
    float p[NSLT];
#pragma unroll
    for (int sl = 0; sl < NSLT; ++sl)
      p[sl] = __hip_atomic_load(as_global(P.partial) + ((size_t)t * NSLT + sl) * N * V + o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int w = NSLT / 2; w > 0; w >>= 1)
#pragma unroll
      for (int i = 0; i < w; ++i) p[i] = p[2 * i] + p[2 * i + 1];
    float v = p[0];
    if ((int)o < P.O) {
      if (P.bias) v += DT::to_float(as_global(P.bias)[o]);
      if (P.out_f32) ((float*)as_global(P.y))[(size_t)t * TP.y_stride + o] = v;
      else ((uint16_t*)as_global(P.y))[(size_t)t * TP.y_stride + o] = DT::from_float(v);
    }
  }
}

// ---- launchers (every part of the build: the file is compiled four times, VPTQ_ST_PART = 1: 2 and 4 token slots + the host
// side, 2: the 8-slot instantiations, 3 / 4: 4 / 8 slots with the reference's roundings - minutes of compile time as one
// translation unit)
template <typename DT, int NSL, bool RES, int V, bool TWO, int TOK, bool EX = false>
static hipError_t launch_st(const SlicedTokGroupParams& P, int grid, uint32_t lds, hipStream_t st) {
  auto kern = gemv_sliced_tok_kernel<DT, NSL, RES, V, TWO, TOK, EX>;
  static std::atomic<bool> attr_set[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (!attr_set[dev]) {
    const hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSLLdsLimit);
    if (e != hipSuccess) return e;
    attr_set[dev] = true;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(kSLThreads), lds, st, P);
  return hipGetLastError();
}
template <typename DT, int TOK>
static hipError_t launch_st_dt(const SlicedTokGroupParams& P, int grid, int v, int nsl, bool res, bool two, uint32_t lds, hipStream_t st) {
  if (v == 16) {
    if (two) return nsl == 16 ? launch_st<DT, 16, false, 16, true, TOK>(P, grid, lds, st) : launch_st<DT, 32, false, 16, true, TOK>(P, grid, lds, st);
    return nsl == 16 ? launch_st<DT, 16, false, 16, false, TOK>(P, grid, lds, st) : launch_st<DT, 32, false, 16, false, TOK>(P, grid, lds, st);
  }
  if (two) return nsl == 8 ? launch_st<DT, 8, false, 8, true, TOK>(P, grid, lds, st) : launch_st<DT, 16, false, 8, true, TOK>(P, grid, lds, st);
  if (nsl == 8) return res ? launch_st<DT, 8, true, 8, false, TOK>(P, grid, lds, st) : launch_st<DT, 8, false, 8, false, TOK>(P, grid, lds, st);
  return res ? launch_st<DT, 16, true, 8, false, TOK>(P, grid, lds, st) : launch_st<DT, 16, false, 8, false, TOK>(P, grid, lds, st);
}

template <typename DT, int TOK>
static hipError_t launch_st_ex(const SlicedTokGroupParams& P, int grid, int v, int nsl, bool res, uint32_t lds, hipStream_t st) {
  if (v == 16) return nsl == 16 ? launch_st<DT, 16, false, 16, false, TOK, true>(P, grid, lds, st) : launch_st<DT, 32, false, 16, false, TOK, true>(P, grid, lds, st);
  if (nsl == 8) return res ? launch_st<DT, 8, true, 8, false, TOK, true>(P, grid, lds, st) : launch_st<DT, 8, false, 8, false, TOK, true>(P, grid, lds, st);
  return res ? launch_st<DT, 16, true, 8, false, TOK, true>(P, grid, lds, st) : launch_st<DT, 16, false, 8, false, TOK, true>(P, grid, lds, st);
}
hipError_t launch_st_ex4(int dtype, const SlicedTokGroupParams& P, int grid, int v, int nsl, bool res, uint32_t lds, hipStream_t st);
hipError_t launch_st_ex8(int dtype, const SlicedTokGroupParams& P, int grid, int v, int nsl, bool res, uint32_t lds, hipStream_t st);
#if !defined(VPTQ_ST_PART) || VPTQ_ST_PART == 3
hipError_t launch_st_ex4(int dtype, const SlicedTokGroupParams& P, int grid, int v, int nsl, bool res, uint32_t lds, hipStream_t st) {
  return dtype == VPTQ_DTYPE_F16 ? launch_st_ex<F16, 4>(P, grid, v, nsl, res, lds, st) : launch_st_ex<BF16, 4>(P, grid, v, nsl, res, lds, st);
}
#endif
#if !defined(VPTQ_ST_PART) || VPTQ_ST_PART == 4
hipError_t launch_st_ex8(int dtype, const SlicedTokGroupParams& P, int grid, int v, int nsl, bool res, uint32_t lds, hipStream_t st) {
  return dtype == VPTQ_DTYPE_F16 ? launch_st_ex<F16, 8>(P, grid, v, nsl, res, lds, st) : launch_st_ex<BF16, 8>(P, grid, v, nsl, res, lds, st);
}
#endif

hipError_t launch_st_tok8(int dtype, const SlicedTokGroupParams& P, int grid, int v, int nsl, bool res, bool two, uint32_t lds, hipStream_t st);
#if !defined(VPTQ_ST_PART) || VPTQ_ST_PART == 2
hipError_t launch_st_tok8(int dtype, const SlicedTokGroupParams& P, int grid, int v, int nsl, bool res, bool two, uint32_t lds, hipStream_t st) {
  return dtype == VPTQ_DTYPE_F16 ? launch_st_dt<F16, 8>(P, grid, v, nsl, res, two, lds, st) : launch_st_dt<BF16, 8>(P, grid, v, nsl, res, two, lds, st);
}
#endif

#if !defined(VPTQ_ST_PART) || VPTQ_ST_PART == 1
// ---- host side -------------------------------------------------------------------
static bool st_exact_ok(const VptqLayerDesc& d) { return !sl_two(d) && gemv_sliced_eligible(d, true); }
static size_t st_partial_bytes(const VptqLayerDesc& d, int tokens, bool exact) {
  const size_t parts = exact ? (size_t)gemv_sliced_slices(d, true) : (size_t)gemv_sliced_slices(d) * (sl_two(d) ? 2 : 1);
  return ((size_t)tokens * parts * d.num_indices * d.vector_len * sizeof(float) + 255) / 256 * 256;
}
static size_t st_counter_bytes(const VptqLayerDesc& d) {
  return (((size_t)(d.num_indices + kSLWaves - 1) / kSLWaves) * sizeof(uint32_t) + 255) / 256 * 256;
}
// ... + (a layer with an input permutation) x[perm] of every token: the kernel stages activations in COLUMN order and a load
// the compiler can see inside its stream loop costs it the counted waits, so the gather is a pre-pass (gemv_k256c.hip:
// permute_x_kernel, one workgroup per 2048 columns and token)
static size_t st_perm_bytes(const VptqLayerDesc& d, int tokens) { return (size_t)tokens * gemv_k256c_perm_bytes(d); }
// (one size for both arithmetics: the exact layouts may have twice the slices)
static size_t st_partial_bytes_max(const VptqLayerDesc& d, int tokens) {
  const size_t a = gemv_sliced_eligible(d) ? st_partial_bytes(d, tokens, false) : 0, b = st_exact_ok(d) ? st_partial_bytes(d, tokens, true) : 0;
  return a > b ? a : b;
}
// ... and, in front of the partial sums (a fixed place whatever the token count), the accumulator words of the ONE-PASS route:
// 2 / 3 tokens in the reference's roundings through the one-token kernel (gemv_sliced.hip, TOK), zero between launches
static size_t st_acc_bytes(const VptqLayerDesc& d) { return gemv_sliced_eligible(d, true) ? gemv_sliced_exact_tokens_workspace_bytes(d, 3) : 0; }
size_t gemv_sliced_tok_workspace_bytes(const VptqLayerDesc& d, int tokens) {
  return st_counter_bytes(d) + st_acc_bytes(d) + st_partial_bytes_max(d, tokens) + st_perm_bytes(d, tokens);
}
// VPTQ_SLICED_ONE_PASS=0: 2 / 3 exact tokens through the column-phase kernel as well (A/B runs)
static bool st_one_pass(const VptqLayerDesc& d, int tokens, bool exact) {
  static std::atomic<int> on{-1};
  if (on < 0) { const char* e = vptq::tune_env("VPTQ_SLICED_ONE_PASS"); on = (e && atoi(e) == 0 && e[0] == '0') ? 0 : 1; }
  return exact && on == 1 && gemv_sliced_exact_tokens_ok(d, tokens);
}

// rows per wave: one round of workgroups (slices x tables x row blocks of 16 waves ~ the CUs)
static int st_rows_per_wave(const VptqLayerDesc& d, bool exact) {
  const long long nslt = exact ? (long long)gemv_sliced_slices(d, true) : (long long)gemv_sliced_slices(d) * (sl_two(d) ? 2 : 1);
  long long r = ((long long)d.num_indices * nslt + kSLWaves * 256 - 1) / (kSLWaves * 256);
  return (int)(r < 1 ? 1 : r > kSLMaxRowsPerWave ? kSLMaxRowsPerWave : r);
}

struct StPlan { int tok, phases, rpw, reg_sums; uint32_t x_off, bd_off, res_off, sum_off, sb_off, lds; };
// the template's token count (2 or 4), the fewest phases whose activations fit beside the table, and the LDS map.  The rows'
// sums (16 waves x rows per wave x tokens x v floats) must fit too: where one round of workgroups does not leave room for
// them even with 4 phases (v = 16 with two tables: 64 floats per row), fewer rows per wave - more workgroups - do.
static bool st_plan(const VptqLayerDesc& d, const VptqSlicedLayout* L, int tokens, bool exact, StPlan& pl, int rpw0 = 0) {
  if (tokens < 2 || tokens > 8) return false;
  const bool res = sl_res256(d), two = sl_two(d);
  if (exact && !st_exact_ok(d)) return false;
  static std::atomic<int> tok4{-1};   // VPTQ_SLICED_TOK4=1: 2 tokens through the 4-slot (matrix-pipe) kernel too (A/B runs)
  if (tok4 < 0) { const char* e = vptq::tune_env("VPTQ_SLICED_TOK4"); tok4 = (e && atoi(e) == 1) ? 1 : 0; }
  pl.tok = tokens > 4 ? 8 : (tokens == 2 && !tok4 && !exact) ? 2 : 4;   // (the reference's roundings: matrix-pipe mode only)
  uint32_t tab = sl_tab_bytes(d, d.num_centroids, 0, exact);
  if (two) {
    const uint32_t t1 = sl_tab_bytes(d, d.num_res_centroids, L[1].whole_table);
    tab = t1 > tab ? t1 : tab;
  }
  pl.x_off = (tab + 15u) & ~15u;
  const int G = d.group_size;
  const int wcols = (G + kSTWindows * 8 - 1) / (kSTWindows * 8) * 8;
  static std::atomic<int> min_phases{-1};   // VPTQ_SLICED_MIN_PHASES=2 / 4: more phases than the LDS asks for (A/B runs)
  if (min_phases < 0) { const char* e = vptq::tune_env("VPTQ_SLICED_MIN_PHASES"); const int v = e ? atoi(e) : 1; min_phases = (v == 2 || v == 4) ? v : 1; }
  static std::atomic<int> no_reg_sums{-1};  // VPTQ_SLICED_LDS_SUMS=1: the rows' sums in LDS in every mode (A/B runs)
  if (no_reg_sums < 0) { const char* e = vptq::tune_env("VPTQ_SLICED_LDS_SUMS"); no_reg_sums = (e && atoi(e) == 1) ? 1 : 0; }
  for (int rpw = rpw0 > 0 ? rpw0 : st_rows_per_wave(d, exact); rpw >= 1; rpw = rpw > 1 ? (rpw + 1) / 2 : 0) {
    for (int phases = min_phases; phases <= kSTWindows; phases *= 2) {
      const int wmax = (kSTWindows / phases) * wcols < G ? (kSTWindows / phases) * wcols : G;   // columns of the widest phase
      uint32_t o = pl.x_off + (uint32_t)(wmax + 8) * (uint32_t)pl.tok * 2u;
      o = (o + 15u) & ~15u;
      pl.sb_off = o; o += exact ? (uint32_t)(wmax + 8) * 8u : 0u;
      pl.bd_off = o; o += (uint32_t)pl.tok * kSLWaves * 4u;
      pl.res_off = o; o += res ? 4096u : 0u;
      const int reg_sums = pl.tok >= 4 && rpw <= kSTRegRows && !no_reg_sums;   // (matrix-pipe mode: 4 registers per row)
      pl.sum_off = o; o += reg_sums ? 0u : (uint32_t)kSLWaves * (uint32_t)rpw * (uint32_t)(pl.tok * d.vector_len) * 4u;
      if (o <= kSLLdsLimit) { pl.phases = phases; pl.rpw = rpw; pl.reg_sums = reg_sums; pl.lds = o; return true; }
    }
  }
  return false;
}

int gemv_sliced_tok_one_pass_parts(const VptqLayerDesc& d, int tokens, bool exact) { return st_one_pass(d, tokens, exact) ? gemv_sliced_exact_tokens_parts(d, tokens) : 0; }
bool gemv_sliced_tok_eligible(const VptqLayerDesc& d, const VptqSlicedLayout* L, int tokens, bool exact) {
  StPlan pl;
  if (!L) return false;
  // (ONE pass of the one-token kernel: also the two-table formats of v = 8, their residual entries gathered once for all tokens)
  // (no column windows needed - unless the columns are taken in window parts)
  if (st_one_pass(d, tokens, exact))
    return L[0].n_slices == gemv_sliced_slices(d, true) && (!sl_two(d) || L[0].res) && (gemv_sliced_exact_tokens_parts(d, tokens) == 1 || L[0].wstart);
  if (!(exact ? st_exact_ok(d) : gemv_sliced_eligible(d))) return false;
  const int n = exact ? 1 : gemv_sliced_tables(d);
  for (int i = 0; i < n; ++i)
    if (!L[i].wstart || L[i].n_slices != gemv_sliced_slices(d, exact)) return false;   // (the arithmetic's own slice count)
  return st_plan(d, L, tokens, exact, pl);
}

// one layer's parameter block (its permutation pre-pass is queued into perm_*: one launch for the whole group)
struct StPermJobs { VptqLayerDesc d[8 * kSTMaxGroup]; const void* xin[8 * kSTMaxGroup]; void* xout[8 * kSTMaxGroup]; int n; };
static hipError_t st_fill(const VptqLayerDesc& d, const VptqSlicedLayout* L, const void* x, void* y, int tokens, int flags, void* ws,
                          int rpw0, SlicedTokParams& TP, StPlan& pl, StPermJobs& jobs) {
  const bool exact = (flags & VPTQ_GEMV_EXACT) != 0;
  if (!gemv_sliced_tok_eligible(d, L, tokens, exact) || !st_plan(d, L, tokens, exact, pl, rpw0)) return hipErrorInvalidValue;
  const bool res = sl_res256(d), two = sl_two(d);
  const int nsl = gemv_sliced_slices(d, exact);
  if (!sl_layout_ok(d, L[0], nsl, res, d.num_centroids) || L[0].whole_table != 0 ||
      (two && (!sl_layout_ok(d, L[1], nsl, false, d.num_res_centroids) || L[1].whole_table != gemv_sliced_whole_table(d, 1))) ||
      !ws || (((uintptr_t)x) & 15) != 0 || (d.in_features % 8) != 0)
    return hipErrorInvalidValue;
  TP = SlicedTokParams{};
  SlicedParams& P = TP.p;
  P.elems = (const uint32_t*)L[0].elems;
  P.res = res ? (const uint8_t*)L[0].res : nullptr;
  P.rcent = res ? (const uint32_t*)d.res_centroids : nullptr;
  P.blocks = (const int32_t*)L[0].blocks;
  P.first = (const int32_t*)L[0].first;
  P.cent = (const uint32_t*)d.centroids;
  P.tab0 = sl_tab_bytes(d, d.num_centroids, 0, exact);
  P.stride0 = P.tab0;
  TP.wstart = (const int32_t*)L[0].wstart;
  if (two) {
    P.elems2 = (const uint32_t*)L[1].elems;
    P.blocks2 = (const int32_t*)L[1].blocks;
    P.first2 = (const int32_t*)L[1].first;
    P.cent2 = (const uint32_t*)d.res_centroids;
    P.tab1 = sl_tab_bytes(d, d.num_res_centroids, L[1].whole_table);
    P.stride1 = L[1].whole_table ? 0u : P.tab1;
    TP.wstart2 = (const int32_t*)L[1].wstart;
  }
  P.x_off = pl.x_off;
  P.x = (const uint16_t*)x;
  P.scale = (const uint16_t*)(d.perm ? d.scale_permuted : d.weight_scale);
  P.wbias = (const uint16_t*)d.weight_bias;
  P.cbias = (const uint16_t*)(d.perm ? d.bias_permuted : d.weight_bias);
  P.perm = nullptr;
  TP.xs = (const uint16_t*)x;
  TP.x_stride = d.in_features;
  if (d.perm) {
    char* const base = (char*)ws + st_counter_bytes(d) + st_acc_bytes(d) + st_partial_bytes_max(d, tokens);
    for (int t = 0; t < tokens; ++t) {
      jobs.d[jobs.n] = d;
      jobs.xin[jobs.n] = (const uint16_t*)x + (size_t)t * d.in_features;
      jobs.xout[jobs.n] = base + (size_t)t * gemv_k256c_perm_bytes(d);
      ++jobs.n;
    }
    TP.xs = (const uint16_t*)base;
    TP.x_stride = (int)(gemv_k256c_perm_bytes(d) / 2);
  }
  P.bias = (const uint16_t*)d.bias;
  // (the arrival counters FIRST: a workspace sized - and zeroed once - for 4 tokens serves 2 and 3 as well)
  P.arrived = (uint32_t*)ws;
  P.partial = (float*)((char*)ws + st_counter_bytes(d) + st_acc_bytes(d));
  P.y = y;
  P.N = d.num_indices; P.G = d.group_size; P.O = d.out_features;
  P.rows_per_wave = pl.rpw;
  const int rows_per_wg = kSLWaves * pl.rpw;
  P.n_rowblocks = (d.num_indices + rows_per_wg - 1) / rows_per_wg;
  P.out_f32 = (flags & VPTQ_GEMV_OUT_F32) ? 1 : 0;
  TP.tokens = tokens;
  TP.phases = pl.phases;
  TP.wcols = (d.group_size + kSTWindows * 8 - 1) / (kSTWindows * 8) * 8;
  TP.x_in_stride = d.in_features;
  TP.y_stride = d.out_features;
  TP.bd_off = pl.bd_off; TP.res_off = pl.res_off; TP.sum_off = pl.sum_off; TP.sb_off = pl.sb_off;
  TP.reg_sums = pl.reg_sums;
  return hipSuccess;
}

// n <= kSTMaxGroup layers of ONE format (vector length, slices, residual kind, dtype) and one input width reading the same x
bool gemv_sliced_tok_groupable(const VptqLayerDesc* d, const VptqSlicedLayout* L, int n, int tokens, bool exact) {
  if (n < 1 || n > kSTMaxGroup || !gemv_sliced_groupable(d, n, exact)) return false;
  const int tables = exact ? 1 : gemv_sliced_tables(d[0]);
  for (int i = 0; i < n; ++i)
    if (!gemv_sliced_tok_eligible(d[i], L + (size_t)i * tables, tokens, exact)) return false;
  return true;
}

// x: [tokens][in_features], y[i]: [tokens][out_features of layer i] (fp32 with VPTQ_GEMV_OUT_F32); ws[i]:
// gemv_sliced_tok_workspace_bytes of layer i, zero before its first use (every launch leaves the counters zero)
hipError_t launch_gemv_sliced_tok_group(const VptqLayerDesc* d, const VptqSlicedLayout* L, int n, const void* x, void* const* y,
                                        int tokens, int flags, void* const* ws, hipStream_t st) {
  const bool exact = (flags & VPTQ_GEMV_EXACT) != 0;
  if (!gemv_sliced_tok_groupable(d, L, n, tokens, exact)) return hipErrorInvalidValue;
  {   // 2 / 3 tokens in the reference's roundings: ONE pass of the one-token kernel where every member's operands fit its LDS
    bool one_pass = true;
    for (int i = 0; i < n; ++i) one_pass = one_pass && st_one_pass(d[i], tokens, exact);
    if (one_pass) {
      void* acc[kSTMaxGroup];
      for (int i = 0; i < n; ++i) acc[i] = (char*)ws[i] + st_counter_bytes(d[i]);
      return launch_gemv_sliced_group(d, L, n, x, y, flags, acc, st, tokens);
    }
    // (column parts of one layer share their output and accumulator words: the one pass only - the column-phase kernel below keeps
    // partial sums and counters per layer)
    if (flags & VPTQ_GEMV_COLUMN_PARTS) return hipErrorInvalidValue;
  }
  SlicedTokGroupParams GP = {};
  GP.n = n;
  StPermJobs jobs = {};
  const int tables = exact ? 1 : gemv_sliced_tables(d[0]);
  const int nslt = gemv_sliced_slices(d[0], exact) * tables;
  // rows per wave: one round of workgroups over ALL members
  long long rows = 0;
  for (int i = 0; i < n; ++i) rows += d[i].num_indices;
  long long r0 = (rows * nslt + kSLWaves * 256 - 1) / (kSLWaves * 256);
  int rpw0 = n == 1 ? 0 : (int)(r0 < 1 ? 1 : r0 > kSLMaxRowsPerWave ? kSLMaxRowsPerWave : r0);
  // (3 - 4 tokens: rather a second round of workgroups than the rows' sums out of the registers - gate / up of an 8B model,
  // 2 x 14336 outputs, v8-k65536-256: 7 rows per wave 55.5 us, 4 rows per wave 52.3 - but not a third and fourth round: the
  // two-table format of the same layers, 14 rows per wave 72.9 us, 4 rows per wave 90.9)
  if (tokens > 2 && rpw0 > kSTRegRows && rpw0 <= 2 * kSTRegRows) rpw0 = kSTRegRows;
  uint32_t lds = 0;
  int tok = 0;
  for (int i = 0; i < n; ++i) {
    StPlan pl;
    const hipError_t e = st_fill(d[i], L + (size_t)i * tables, x, y[i], tokens, flags, ws[i], rpw0, GP.p[i], pl, jobs);
    if (e != hipSuccess) return e;
    lds = pl.lds > lds ? pl.lds : lds;
    tok = pl.tok;
    GP.start[i + 1] = GP.start[i] + nslt * GP.p[i].p.n_rowblocks;
  }
  for (int i = n; i < kSTMaxGroup; ++i) GP.start[i + 1] = GP.start[n];
  if (jobs.n > 0) {
    const hipError_t e = launch_permute_x(jobs.d, jobs.n, jobs.xin, jobs.xout, st);
    if (e != hipSuccess) return e;
  }
  const int grid = GP.start[n];
  const bool res = sl_res256(d[0]), two = sl_two(d[0]);
  const int nsl = gemv_sliced_slices(d[0], exact);
  if (exact)
    return tok == 8 ? launch_st_ex8(d[0].dtype, GP, grid, d[0].vector_len, nsl, res, lds, st)
                    : launch_st_ex4(d[0].dtype, GP, grid, d[0].vector_len, nsl, res, lds, st);
  if (tok == 8) return launch_st_tok8(d[0].dtype, GP, grid, d[0].vector_len, nsl, res, two, lds, st);
  if (d[0].dtype == VPTQ_DTYPE_F16)
    return tok == 2 ? launch_st_dt<F16, 2>(GP, grid, d[0].vector_len, nsl, res, two, lds, st)
                    : launch_st_dt<F16, 4>(GP, grid, d[0].vector_len, nsl, res, two, lds, st);
  return tok == 2 ? launch_st_dt<BF16, 2>(GP, grid, d[0].vector_len, nsl, res, two, lds, st)
                  : launch_st_dt<BF16, 4>(GP, grid, d[0].vector_len, nsl, res, two, lds, st);
}
hipError_t launch_gemv_sliced_tok(const VptqLayerDesc& d, const VptqSlicedLayout* L, const void* x, void* y, int tokens, int flags,
                                  void* ws, hipStream_t st) {
  void* const ys[1] = {y};
  void* const wss[1] = {ws};
  return launch_gemv_sliced_tok_group(&d, L, 1, x, ys, tokens, flags, wss, st);
}

#endif   // part 1

}  // namespace vptq
