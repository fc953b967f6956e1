// Fused dequant + GEMV for the large-codebook formats of the published checkpoints - vector length 8 or 16, 16384 ...
// 65536 main centroids, any residual codebook: "v8-k65536-0" (T = 16), "v8-k65536-256" (T = 24: most checkpoints),
// "v8-k65536-65536" (T = 32, the "4 bit" format), "v16-k65536-65536", "v16-k65536-1024", "v8-k32768-0", ... - over LOAD-TIME
// DERIVED LAYOUTS that make every centroid gather LDS-local.  A residual codebook other than v8's 256-entry one is a second
// table with a layout of its own, served by its own workgroups in the same launch: (c + r) s x = c s x + r s x.
// Same contract as gemv_gather.hip (reference: WqA16WithOutliers_PackIndice,
// csrc/kernels/quant_gemv.cuh:11-186, dispatch csrc/quant_gemv.cu:54-132), one token.
//
// Why.  A 1 MiB codebook cannot live in LDS, so gemv_gather.hip gathers centroid rows from L2 with one
// 16-byte load per index: every gather moves a 128-byte line for 16 useful bytes and the launch is
// bound by the L2 -> L1 fill rate (39.6 us per 8192^2 layer, 0.06 of the HBM roofline).
// Here the codebook is cut into 8 SLICES of 8192 entries (128 KiB of a workgroup's LDS; 16 slices of 4096 for
// layers whose activations do not fit beside that: more than 14336 / 14080 columns) and, once per
// layer (vptq_amd/utils/sliced.py; the on-disk tensors stay the state-dict contract), every row's
// elements are bucketed by the slice of their index:
//   elems  : for slice s, for row n = 0..N-1: the row's elements whose index lies in slice s, padded to a
//            multiple of 64 with (column = G, local index = 0); one 32-bit word per element =
//            column | (index mod slice size) << 16; inside a (s, n) list the order is free - the builder picks
//            one in which 16 consecutive elements hit different LDS bank groups
//   blocks : [slices][N] number of 64-element blocks of (s, n);  first : [slices][N] index of its first block
//   res    : (residual formats) one byte per element, same order and padding: its residual index; the 256-entry
//            residual codebook sits in LDS beside the slice
// = 4 instead of 2 bytes per element for T = 16, 5 instead of 3 for T = 24 (+ ~3 % padding): the layout costs
// 2.06x / 1.72x the packed indices in memory on top of them and in HBM traffic per token.  A workgroup owns
// (slice, block of rows): it copies its slice (LDS-DMA) and f16(scale * x) of all columns into LDS, then each wave
// streams the CONTIGUOUS element blocks of its consecutive rows through a register queue; per element one
// ds_read_b128 (entry) + one ds_read_u16 (activation) + 8 FMAs in fp32.  A row's partial sum per slice is added to the
// output's 64-bit accumulator word in the caller's workspace (fixed point + arrival count, one returning atomic); the lane
// whose add completes the count rounds the total once, adds the output bias and stores y.  Two arithmetics:
//   folded (flags = 0):   y = sum (c + r)[idx] * f16(s x) + sum b x + bias   (sum b x rides with slice 0; two-table formats:
//                         c f16(s x) and r f16(s x) from different workgroups)
//   reference roundings (VPTQ_GEMV_EXACT, template EX): w = f16(f16(f16(c + r) * s) + b) per weight, y = sum w x + bias -
//                         the product default since round 5; one table only (none or the 256-entry residual codebook)
#include <cstring>

#include "sliced.h"

namespace vptq {

// element blocks in flight per wave.  Same-box A/B (profiles/r03/sliced_queue_ab.txt): depth 8 / 16 / 32 = 14.2 /
// 16.2 / 21.2 us per 8192^2 layer - the unrolled loop runs ceil(blocks / depth) * depth steps, and a wave's stream is
// only ~34 blocks long: the steps past its end cost as much as real ones (8- and 16-byte loads per lane made no
// difference: it is the steps, not the load instructions or the bytes).
#ifndef VPTQ_SLICED_QUEUE
#define VPTQ_SLICED_QUEUE 8
#endif
constexpr int kSLQueueWords = VPTQ_SLICED_QUEUE;
// timing-only ablations (results wrong): bit 0 no gathers / FMAs, bit 1 no slice copy, bit 2 no activation staging
#ifndef VPTQ_SLICED_ABLATE
#define VPTQ_SLICED_ABLATE 0
#endif
// Up to kSLMaxGroup layers that read the SAME activation (q / k / v, gate / up) in one launch: layer l owns the workgroups
// [start[l], start[l + 1]).  One launch instead of n: the fixed part of a launch (boundary, slice copy, staging, the
// cross-slice hand-over: ~7 of the 10 us of a 4096 x 4096 layer) is paid once.
constexpr int kSLMaxGroup = 3;
// How the slices of a row meet (round 5): ONE hop.  Every (slice, output) partial sum becomes a 64-bit fixed-point word
// (count | value) that is added to the output's accumulator word with a returning device-scope atomic; integer adds commute, so
// the sum does not depend on the order, and the lane whose add completes the count holds the total, rounds it once and stores y.
// (Rounds 3-4: write-through partial sums, an arrival counter per row block and a read-back by the last workgroup - three
// dependent trips through memory, ~4 us measured as a launch of its own; the phase stamps of round 5 show that in ONE launch
// the stragglers of the stream hid most of it: the two forms time the same, profiles/r05/sliced_epilogue_ab.txt.  The one-hop
// form stays: no counters, no barriers, no read-back, half the workspace.)
// phase time stamps (tools/sliced_trace.py): every wave writes s_memrealtime at entry / after the prologue barrier / at the
// end of its stream / at its exit behind the accumulator words of the workspace (8192^2-sized layers only: the room the
// round-4 partial sums had)
#ifndef VPTQ_SLICED_TRACE
#define VPTQ_SLICED_TRACE 0
#endif
// A/B: issue priority by wave age (the phase stamps show the 4 waves of a SIMD finishing 1.3 us apart, oldest first):
// 1 = the youngest wave of a SIMD gets the highest priority, 2 = the oldest
#ifndef VPTQ_SLICED_PRIO
#define VPTQ_SLICED_PRIO 0
#endif
// 1: a block's LDS gathers are issued one step ahead of its arithmetic (two register sets); 0: gather, wait, compute per step
// RG: blocks per queue stage (A/B)
// (measured, profiles/r05/sliced_exact_two_table_queue_ab.txt: 8 / 4 / 2 blocks = 65.8 / 61.2 / 60.1 us at 8192^2 v8-k65536-65536 - the
// L1 miss path of a CU is the limit, more gathers in flight only queue up in front of the element words)
#ifndef VPTQ_SLICED_RGQ
#define VPTQ_SLICED_RGQ 2
#endif
#ifndef VPTQ_SLICED_RGNT
#define VPTQ_SLICED_RGNT 0
#endif
#ifndef VPTQ_SLICED_PIPE
#define VPTQ_SLICED_PIPE 1
#endif
// accumulator word of one output: bits [0, 7) arrivals, [7, 14) arrivals whose partial sum could not be represented (NaN, infinite,
// beyond the field), [14, 64) the sum in units of 2^-F as a 50-bit two's-complement number.  F = 30 for fp16 layers: partial sums up
// to 2^17 (twice the type's range), resolution 9.3e-10 - a 16-bit output of magnitude 1e-4 still gets its sum to 1e-5 relative per
// arrival; F = 28 for bf16 layers: partial sums up to 2^19 = 5.2e5 (no activation of a language model comes near; the type itself
// goes to 3e38), resolution 3.7e-9 against an ulp of 4.8e-7 at 1e-4.  Truncation is towards zero (magnitude first, then the sign):
// no bias.  Integer adds commute and wrap: the result is the same whoever arrives last, and a transient overflow of the field does
// not matter as long as the final sum fits it (2 bits above the partial sums' limit).  A partial sum that cannot be represented
// makes the OUTPUT NaN - loud - instead of saturating to a finite value.
// [Round 5: units of 2^-24 for every type, floor (a bias of up to 6e-8 per arrival, always downwards: 16 - 96 arrivals moved
// outputs of magnitude 1e-3 by several fp16 ulps), clamp to +-3.3e7 - a finite bf16 value.  ADVICE r5.]
constexpr int kSLFixShift = 14;
template <typename DT> constexpr int sl_frac() { return std::is_same<DT, F16>::value ? 30 : 28; }
template <int F>
static __device__ __forceinline__ unsigned long long sl_to_fixed(float v) {
  // |v| 2^F as a 64-bit integer out of two 32-bit conversions (the compiler's float -> int64 is ~25 instructions, and a wave pays
  // it at the end of every row): |v| = a + f with a = floor(|v|) < 2^25 and f = |v| - a in [0, 1), exact in fp32
  constexpr float lim = (float)(1u << (49 - F - 2));   // fp16: 2^17; bf16: 2^19
  const float m = __builtin_fabsf(v);
  const bool bad = !(m < lim);                       // NaN, infinite, beyond the field
  const float c = bad ? 0.f : m;
  const float a = __builtin_floorf(c);
  const uint32_t ai = (uint32_t)a;
  const uint32_t fi = (uint32_t)((c - a) * (float)(1u << F));   // < 2^F: f <= 1 - 2^-24
  unsigned long long q = ((unsigned long long)ai << F) + (unsigned long long)fi;
  if (v < 0.f) q = 0ull - q;
  return (q << kSLFixShift) + (bad ? 129ull : 1ull);
}
template <int F>
static __device__ __forceinline__ float sl_from_fixed(unsigned long long w) {
  if ((w >> 7) & 127ull) return __builtin_nanf("");
  const long long q = (long long)w >> kSLFixShift;
  return (float)((double)q * (1.0 / (double)(1ull << F)));   // |q| < 2^49: exact in fp64, ONE rounding to fp32
}
struct SlicedGroupParams {
  int n;
  int arrivals;   // workgroups that add into an output's accumulator word: (tables x) slices - x n where the "layers" are COLUMN PARTS of one
  int start[kSLMaxGroup + 1];
  SlicedParams p[kSLMaxGroup];
};

// EX (round 5): the reference's roundings per weight instead of the folded form - w = f16(f16(f16(c + r) * s) + b)
// (vptq/ops/quant_gemm.py:121,155-156), y = sum w x in fp32: what gemv_gather computes, here with LDS-local gathers.  The
// column's scale and bias are staged as a word per column beside the raw activations (6 instead of 2 bytes per column: the
// host picks 16 slices where 8 no longer fit); one table only - c and r must meet in one lane.
// RG (EX only): any OTHER residual codebook (2 ... 65536 entries: the "4 bit" v8-k65536-65536, the "2 bit" v16-k65536-65536, ...) in
// the reference's roundings - c and r must meet in one lane, and a second 1 - 2 MiB table cannot sit in LDS as well: the main
// entry comes out of the LDS slice, the residual entry is GATHERED FROM L2 by its index (a uint16 side stream of the layout), a
// second queue stage ahead of the arithmetic.  One of the gather kernel's two cache gathers per element instead of both.
// TOK = 2 / 3 (EX, one table): that many TOKENS in one pass over the layout - the weight is rebuilt once per element and meets
// every token's activation (2 / 4 more multiply-adds per pair of outputs; the stream, the gathers and the three roundings are
// paid once).  x [TOK][x_stride], y [TOK][y_stride], accumulator words [TOK][acc_stride]; LDS: tokens 0 and 1 interleaved per
// column (one 32-bit gather), token 2 a plane of its own: 2 TOK + 4 bytes per column - layers whose exact layout has 16 (v = 16:
// 32) slices, up to ~12000 (2 tokens) / ~9700 (3) columns.  More tokens: gemv_sliced_tok.hip (column phases, matrix pipe).
// WPT (TOK > 1): WINDOW PARTS - where the slice leaves no room for (2 tokens + 4) bytes of EVERY column (4096 columns beside a 128 KiB
// slice, 14336 beside 64 KiB), a workgroup takes only some of the layout's four column windows (P.wparts = 2 or 4 workgroups per
// (slice, row block)): it stages those columns alone and walks, row by row, the blocks that hold its windows' part of the list
// (`wstart`; blocks do not end where windows do - an element of another window reads the zero column, as the padding does).  The
// parts meet in the outputs' accumulator words like slices do: slices x parts arrivals.
template <typename DT, int NSL, bool RES, int V = 8, bool TWO = false, bool EX = false, bool RG = false, int TOK = 1, bool WPT = false>
__global__ __launch_bounds__(kSLThreads) void gemv_sliced_kernel(const SlicedGroupParams GP) {
  // this workgroup's layer; its parameters come out of the kernel-argument segment through the scalar cache (a run-time
  // index into the by-value argument would make the compiler copy it to scratch memory)
  int layer = 0;
  if (GP.n > 1 && (int)blockIdx.x >= GP.start[1]) layer = 1;
  if (GP.n > 2 && (int)blockIdx.x >= GP.start[2]) layer = 2;
  layer = __builtin_amdgcn_readfirstlane(layer);
  const int bx = (int)blockIdx.x - (layer == 0 ? 0 : layer == 1 ? GP.start[1] : GP.start[2]);
#if defined(__HIP_DEVICE_COMPILE__)
  typedef const char __attribute__((address_space(4)))* sl_kernarg_t;
  typedef const SlicedParams __attribute__((address_space(4)))* sl_params_t;
  const SlicedParams P = *(sl_params_t)((sl_kernarg_t)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(SlicedGroupParams, p) +
                                         (size_t)layer * sizeof(SlicedParams));
#else
  const SlicedParams P = GP.p[0];   // (host pass of the compiler: never executed)
#endif
  static_assert(((V == 8 && (NSL == 8 || NSL == 16)) || (V == 16 && (NSL == 16 || NSL == 32))) && (V == 8 || !RES) && !(TWO && RES) &&
                !(EX && TWO) && (!RG || (EX && !RES && !TWO)) && (TOK == 1 || ((TOK == 2 || TOK == 3) && EX)) && (!WPT || (TOK > 1 && !RG)), "slices");
  constexpr int NSLT = TWO ? 2 * NSL : NSL;   // workgroups per row block: one per (table, slice)
  constexpr int EPL = 1;   // element words per lane and block (2 and 4 - 8 / 16-byte loads - were measured: no difference)
  constexpr uint32_t kEntry = V * 2u;                          // bytes of a codebook entry
  const uint32_t kSLXOff = P.x_off;                          // staged activations: (G + 64) halves, behind the table
  typedef uint32_t evec_t __attribute__((ext_vector_type(EPL)));
  constexpr int kLoadsPerStep = (RES || RG) ? 2 : 1;
  // blocks in flight per wave (RG, v = 16: 4 - the gathered residual entries of the second stage are 8 registers each)
  constexpr int kSLQueue = RG ? (V == 16 && VPTQ_SLICED_RGQ > 4 ? 4 : VPTQ_SLICED_RGQ) : (kSLQueueWords / EPL < 2 ? 2 : kSLQueueWords / EPL);
  constexpr int W4 = V / 8;   // 16-byte pieces of an entry
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  {
    typedef __attribute__((address_space(3))) unsigned char lds_u8_t;
    if ((uint32_t)(uintptr_t)(lds_u8_t*)smem != 0u) __builtin_trap();  // absolute LDS addressing
  }
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // (WPT: the window part varies fastest, then the slice)
  const int wparts = WPT ? P.wparts : 1;
  const int wp = WPT ? bx & (wparts - 1) : 0;
  const int bxs = WPT ? bx / wparts : bx;
  const int sg = bxs & (NSLT - 1), rb = bxs / NSLT;
  const int s = sg & (NSL - 1);
  const bool second = TWO && sg >= NSL;   // (uniform over the workgroup) this workgroup gathers from the residual table
  const uint32_t* const elems_t = second ? P.elems2 : P.elems;
  const int32_t* const blocks_t = second ? P.blocks2 : P.blocks;
  const int32_t* const first_t = second ? P.first2 : P.first;
  const uint32_t* const cent_t = second ? P.cent2 : P.cent;
  const int N = P.N;
  // WPT: this workgroup's columns [c0, c0 + G) = its windows; G = what is staged (the other kernels: all columns, c0 = 0)
  const int wper = WPT ? kSLWindows / wparts : 0;                                 // layout windows per part
  const int c0 = WPT ? (wp * wper * P.wcols < P.G ? wp * wper * P.wcols : P.G) : 0;
  const int c1 = WPT ? ((wp == wparts - 1 || (wp + 1) * wper * P.wcols > P.G) ? P.G : (wp + 1) * wper * P.wcols) : P.G;
  const int G = c1 - c0;
  const int GS = WPT ? P.wstage : P.G;                                            // columns the LDS map is laid out for
  const int rpw = P.rows_per_wave;
  const int row0 = (rb * kSLWaves + wave) * rpw;   // this wave's first row
  const int n_rows = row0 >= N ? 0 : (N - row0 < rpw ? N - row0 : rpw);

#if VPTQ_SLICED_TRACE
  unsigned long long* const trace = (unsigned long long*)as_global(P.partial) + (size_t)N * V + ((size_t)blockIdx.x * kSLWaves + wave) * 4;
  if (lane == 0) trace[0] = __builtin_amdgcn_s_memrealtime();
#endif
  // ---- prologue, ordered by what has to be in flight first (round 5; phase stamps of the round-4 order - profiles/r05/
  // sliced_trace_*.txt: the barrier fell at 3.1 us of a 12.6 us launch, the first element words arrived after it: every wave
  // first WAITED ~1 us for its two index words from cold memory, then issued the table copy, waited for it before it staged
  // the activations and only then asked for elements).  vmcnt returns in order, so a wait for a load also waits for everything
  // issued before it: (1) the table copy (LDS-DMA: needs kernel arguments only) goes out first; (2) the staging loads and the
  // rows' block counts follow, nobody waits for them yet; (3) where the wave's stream starts and how long it is come through
  // the SCALAR cache (lgkmcnt, not vmcnt: waiting for them waits for nothing else); (4) the element queue is requested;
  // (5) only then the staging arithmetic, whose wait is "all but the element words" and covers the table copy.
  // (1) this workgroup's part of its table into LDS by LDS-DMA: 1 KiB per instruction and wave (no registers); the last
  // piece of a small table is partial
  {
    const uint32_t tab = second ? P.tab1 : P.tab0;   // (scalar fields: a run-time index into the by-value argument makes the compiler copy it to scratch)
    const uint64_t va = (uint64_t)(uintptr_t)as_global(cent_t) + (uint64_t)s * (second ? P.stride1 : P.stride0) + (uint64_t)lane * 16u;
    for (uint32_t off = (uint32_t)wave * 1024u; off < ((VPTQ_SLICED_ABLATE & 2) ? 0u : tab); off += kSLWaves * 1024u) {
      const uint32_t d = (uint32_t)__builtin_amdgcn_readfirstlane((int)off);
      if (off + (uint32_t)lane * 16u < tab) {
        const uint64_t v = va + (uint64_t)off;
        uint32_t keep_m0;   // (M0 belongs to the compiler: saved and restored inside the statement)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep_m0) : "v"(v), "s"(d) : "memory");
      }
    }
  }
  // residual codebook (256 entries = 4 KiB) behind the activations: waves 0-3 bring 1 KiB each
  // LDS behind the table: [G + 64 halves: f16(s x), EX: x] [EX: G + 64 words {s, b}] [16 floats: sum b x parts] [RES: 4 KiB]
  // (TOK > 1: [G + 64 words: tokens 0 | 1] [TOK = 3: G + 64 halves: token 2] in front of the {s, b} words)
  [[maybe_unused]] const uint32_t x2_off = kSLXOff + (uint32_t)(GS + 64) * 4u;
  const uint32_t sb_off = kSLXOff + (uint32_t)(GS + 64) * 2u * (uint32_t)TOK;    // (EX) scale | bias << 16 per column
  const uint32_t bd_off = sb_off + (EX ? (uint32_t)(GS + 64) * 4u : 0u);         // 16 floats behind the staged operands
  const uint32_t res_off = bd_off + 64u;
  if constexpr (RES) {
    if (wave < 4) {
      const uint64_t v = (uint64_t)(uintptr_t)as_global(P.rcent) + (uint64_t)wave * 1024u + (uint64_t)lane * 16u;
      const uint32_t d = (uint32_t)__builtin_amdgcn_readfirstlane((int)(res_off + (uint32_t)wave * 1024u));
      uint32_t keep_m0;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep_m0) : "v"(v), "s"(d) : "memory");
    }
  }
  // (2) staging loads of the first kSLPre rounds of the staging loop (16376 columns: every layer of the published
  // families); a permutation makes the activation loads depend on its own load (layers that keep one pay that wait here)
  const int chunks = G >> 3;
  // (column-order tensors from this workgroup's first column on: c0 = 0 unless WPT)
  const uint16_t* const scale_c = as_global(P.scale) + c0;
  const uint16_t* const xcol = as_global(P.x) + c0;
  const uint16_t* const perm_c = P.perm != nullptr ? as_global(P.perm) + c0 : nullptr;
  const uint16_t* const cbias_c = EX ? as_global(P.cbias) + c0 : nullptr;
  constexpr int kSLPre = 2;
  struct XT { u32x4 v[TOK]; };   // a chunk's activations, one vector per token ([0] = the permutation's words where there is one)
  XT st_x[kSLPre];
  u32x4 st_s[kSLPre], st_b[kSLPre];
#pragma unroll
  for (int r = 0; r < kSLPre; ++r) {
    const int q = tid + r * kSLThreads;
    st_s[r] = st_b[r] = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
    for (int t = 0; t < TOK; ++t) st_x[r].v[t] = u32x4{0u, 0u, 0u, 0u};
    if (q < chunks && !(VPTQ_SLICED_ABLATE & 4)) {
      st_s[r] = *(const u32x4*)(scale_c + 8 * q);
      if (P.perm == nullptr) {
#pragma unroll
        for (int t = 0; t < TOK; ++t) st_x[r].v[t] = *(const u32x4*)(xcol + (size_t)t * P.x_stride + 8 * q);
      }
      else st_x[r].v[0] = *(const u32x4*)(perm_c + 8 * q);   // (the permutation's words: resolved in (5))
      if constexpr (EX) st_b[r] = *(const u32x4*)(cbias_c + 8 * q);
      else if (sg == 0 && P.wbias != nullptr) st_b[r] = *(const u32x4*)(as_global(P.wbias) + 8 * q);
    }
  }
  // ... and the rows' block counts (lane i: blocks of row row0 + i; read after the barrier)
  int my_blocks = 0;
  [[maybe_unused]] int seg_first = 0;   // (WPT) lane i: the first block of row row0 + i that holds elements of this workgroup's windows
  if constexpr (WPT) {
    // the windows' part of a (slice, row) list = positions [wstart[w0], wstart[w1]) of it: the blocks that overlap them
    if (n_rows > 0 && lane < n_rows) {
      const size_t li = (size_t)s * N + row0 + lane;
      const int32_t* const wsp = as_global(P.wstart) + li * (kSLWindows + 1);
      const int ws = wsp[wp * wper], we = wsp[(wp + 1) * wper];
      my_blocks = we > ws ? ((we + 63) >> 6) - (ws >> 6) : 0;
      seg_first = as_global(first_t)[li] + (ws >> 6);
    }
  } else {
    if (n_rows > 0 && lane < n_rows) my_blocks = (as_global(blocks_t) + (size_t)s * N + row0)[lane];
  }
  // activations: f16(scale * x) of every column, zero for the padding column G; the workgroups of slice 0 also form
  // sum b x (it rides in their partial sums)
  typedef __attribute__((address_space(3))) u32x4 lds_q_t;
  float bd = 0.f;
  auto stage = [&](int q, const XT xin, const u32x4 sv, const u32x4 bv, bool have_perm_words) __attribute__((always_inline)) {
    u32x4 v = {0u, 0u, 0u, 0u};
    [[maybe_unused]] u32x4 vt[TOK > 1 ? TOK : 1];   // (TOK > 1) tokens 1 ..: v is token 0
#pragma unroll
    for (int t = 0; t < (TOK > 1 ? TOK : 1); ++t) vt[t] = u32x4{0u, 0u, 0u, 0u};
    if (q < chunks && !(VPTQ_SLICED_ABLATE & 4)) {
      // the staged operand is in COLUMN order: column c multiplies feature perm[c] (scale in column order comes
      // with the descriptor: scale_permuted); sum b x is taken in input-FEATURE order (a permutation only reorders it)
      u32x4 xv = xin.v[0];
      u32x4 xc = xv;
      if (have_perm_words) {
        const u32x4 pv = xv;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const uint32_t lo = as_global(P.x)[pv[i] & 0xffffu], hi = as_global(P.x)[pv[i] >> 16];
          xc[i] = lo | (hi << 16);
        }
        if constexpr (TOK > 1) {
#pragma unroll
          for (int t = 1; t < TOK; ++t) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const uint32_t lo = (as_global(P.x) + (size_t)t * P.x_stride)[pv[i] & 0xffffu], hi = (as_global(P.x) + (size_t)t * P.x_stride)[pv[i] >> 16];
              vt[t][i] = lo | (hi << 16);
            }
          }
        } else {
          xv = *(const u32x4*)(as_global(P.x) + 8 * q);
        }
      } else if constexpr (TOK > 1) {
#pragma unroll
        for (int t = 1; t < TOK; ++t) vt[t] = xin.v[t];
      }
      if constexpr (EX) {
        v = xc;   // raw activations; the column's scale and bias as one word: {s, b}
        u32x4 w0, w1;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          w0[2 * i] = (sv[i] & 0xffffu) | (bv[i] << 16);
          w0[2 * i + 1] = (sv[i] >> 16) | (bv[i] & 0xffff0000u);
          w1[2 * i] = (sv[2 + i] & 0xffffu) | (bv[2 + i] << 16);
          w1[2 * i + 1] = (sv[2 + i] >> 16) | (bv[2 + i] & 0xffff0000u);
        }
        *(lds_q_t*)(uintptr_t)(sb_off + (uint32_t)q * 32u) = w0;
        *(lds_q_t*)(uintptr_t)(sb_off + (uint32_t)q * 32u + 16u) = w1;
      } else {
        if (sg == 0 && P.wbias != nullptr) {
#pragma unroll
          for (int i = 0; i < 4; ++i) bd = DT::dot2(xv[i], bv[i], bd);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = DT::mul2(xc[i], sv[i]);
      }
    } else if constexpr (EX) {   // the padding column: scale = bias = 0 (and x = 0: w x = 0)
      const u32x4 z = {0u, 0u, 0u, 0u};
      *(lds_q_t*)(uintptr_t)(sb_off + (uint32_t)q * 32u) = z;
      *(lds_q_t*)(uintptr_t)(sb_off + (uint32_t)q * 32u + 16u) = z;
    }
    if constexpr (TOK == 1) {
      *(lds_q_t*)(uintptr_t)(kSLXOff + (uint32_t)q * 16u) = v;
    } else {   // column j of the chunk: {token 0, token 1} in one word; token 2 in its plane
      u32x4 w0, w1;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        w0[2 * i] = (v[i] & 0xffffu) | (vt[1][i] << 16);
        w0[2 * i + 1] = (v[i] >> 16) | (vt[1][i] & 0xffff0000u);
        w1[2 * i] = (v[2 + i] & 0xffffu) | (vt[1][2 + i] << 16);
        w1[2 * i + 1] = (v[2 + i] >> 16) | (vt[1][2 + i] & 0xffff0000u);
      }
      *(lds_q_t*)(uintptr_t)(kSLXOff + (uint32_t)q * 32u) = w0;
      *(lds_q_t*)(uintptr_t)(kSLXOff + (uint32_t)q * 32u + 16u) = w1;
      if constexpr (TOK == 3) *(lds_q_t*)(uintptr_t)(x2_off + (uint32_t)q * 16u) = vt[2];
    }
  };
  // layers of more than 16376 columns: the staging rounds behind the first kSLPre, done HERE - in front of the element
  // queue (a load issued behind it would make every later wait drain the queue; the compiler did exactly that for the
  // slice-0 workgroups' LDS read behind the barrier, found in the ISA)
  for (int q = tid + kSLPre * kSLThreads; q < chunks + 8; q += kSLThreads) {
    u32x4 sv = {0u, 0u, 0u, 0u}, bv = sv;
    XT xv;
#pragma unroll
    for (int t = 0; t < TOK; ++t) xv.v[t] = u32x4{0u, 0u, 0u, 0u};
    if (q < chunks && !(VPTQ_SLICED_ABLATE & 4)) {
      sv = *(const u32x4*)(scale_c + 8 * q);
      xv.v[0] = *(const u32x4*)((P.perm != nullptr ? perm_c : xcol) + 8 * q);
      if constexpr (TOK > 1) {
        if (P.perm == nullptr) {
#pragma unroll
          for (int t = 1; t < TOK; ++t) xv.v[t] = *(const u32x4*)(xcol + (size_t)t * P.x_stride + 8 * q);
        }
      }
      if constexpr (EX) bv = *(const u32x4*)(cbias_c + 8 * q);
      else if (sg == 0 && P.wbias != nullptr) bv = *(const u32x4*)(as_global(P.wbias) + 8 * q);
    }
    stage(q, xv, sv, bv, P.perm != nullptr);
  }
  // (3) this wave's stream: the blocks of its rows are contiguous in `elems`; `first` is a running sum over (slice, row),
  // so its next entry behind the wave's rows ends the stream (the last wave of the last slice has no such entry: it adds its
  // block counts up)
  int total = 0, first_block = 0;
  if constexpr (WPT) {   // (the rows' parts are not contiguous: the stream is walked row by row, its length is the sum)
    int t = my_blocks;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
    total = __builtin_amdgcn_readfirstlane(t);
  } else if (n_rows > 0) {
    typedef const int32_t __attribute__((address_space(4)))* sl_const_i32_t;   // uniform + constant address space = scalar loads
    const size_t at = (size_t)s * N + row0;
    const sl_const_i32_t fp = (sl_const_i32_t)(uintptr_t)as_global(first_t);
    first_block = fp[at];
    if (at + (size_t)n_rows < (size_t)NSL * N) {
      total = fp[at + n_rows] - first_block;
    } else {
      int t = my_blocks;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
      total = __builtin_amdgcn_readfirstlane(t);
    }
  }

  // (4) element queue: block k of the stream -> slot k % kSLQueue
  evec_t eq[kSLQueue];
  uint32_t rq[(RES || RG) ? kSLQueue : 1];
  // (a wave without blocks still issues its counted loads: of block 0 of the layout, which always exists - behind the last
  // list there is nothing to read, and a residual index picked up there would send the second stage's gather anywhere:
  // found by tools/gpu_fuzz.py --sliced with every element in one slice, round 5)
  const size_t fb = (!WPT && total > 0) ? (size_t)first_block : 0;   // (WPT: absolute block indices)
  // (wave-uniform base + the lane's 32-bit byte offset: the scalar-base addressing form - no 64-bit vector add per load)
  const char* const ep = (const char*)(as_global(elems_t) + fb * (64 * EPL));
  const char* const rp = (RES || RG) ? (const char*)as_global(P.res) + fb * 64 * (RG ? 2 : 1) : nullptr;
  const uint32_t lane4 = (uint32_t)lane * 4u * EPL, lane_r = (uint32_t)lane * (RG ? 2u : 1u);
  const int last = total > 0 ? total - 1 : 0;
  int i_next = 0;
  // (past the end of the stream a step still issues its load - every step the same instructions, so the waits
  // stay counted - of the last block again; one cached word for all lanes instead was measured slower)
  // WPT: the issue side's place in the stream - row, blocks left in its part, block (x 64) - as scalars; past the end the last block
  // again (stride 0), as above
  [[maybe_unused]] int iq_row = 0, iq_left = 0x7fffffff, iq_stride = 0;
  [[maybe_unused]] size_t iq_b64 = 0;
  [[maybe_unused]] auto iq_find = [&]() __attribute__((always_inline)) {   // the next row (from iq_row on) with blocks
    while (iq_row < n_rows) {
      const int c = __builtin_amdgcn_readlane(my_blocks, iq_row);
      if (c != 0) {
        iq_left = c;
        iq_b64 = (size_t)__builtin_amdgcn_readlane(seg_first, iq_row) * 64;
        iq_stride = 64;
        return;
      }
      ++iq_row;
    }
    iq_left = 0x7fffffff;
    iq_stride = 0;
  };
  if constexpr (WPT) iq_find();
  auto issue = [&](auto slot_c) __attribute__((always_inline)) {
    constexpr int S = decltype(slot_c)::value;
    if constexpr (WPT) {
      eq[S] = __builtin_nontemporal_load((const evec_t*)(ep + iq_b64 * (4 * EPL) + (size_t)lane4));
      if constexpr (RES) rq[S] = *(const uint8_t*)(rp + iq_b64 + (size_t)lane_r);
      if (--iq_left == 0) {
        ++iq_row;
        iq_find();
      } else {
        iq_b64 += (size_t)iq_stride;
      }
      return;
    }
    const size_t b64 = (size_t)(i_next < last ? i_next : last) * 64;
    eq[S] = __builtin_nontemporal_load((const evec_t*)(ep + b64 * (4 * EPL) + (size_t)lane4));
    if constexpr (RES) rq[S] = *(const uint8_t*)(rp + b64 + (size_t)lane_r);
    if constexpr (RG) rq[S] = *(const uint16_t*)(rp + b64 * 2 + (size_t)lane_r);
    ++i_next;
  };
  sl_for_slots<kSLQueue>([&](auto slot_c) {
    issue(slot_c);
    __builtin_amdgcn_sched_barrier(0);
  });

  // (5) the hoisted staging rounds: arithmetic + LDS stores (their loads are older than the element words)
  {
#pragma unroll
    for (int r = 0; r < kSLPre; ++r) {
      const int q = tid + r * kSLThreads;
      if (q < chunks + 8) stage(q, st_x[r], st_s[r], st_b[r], P.perm != nullptr);
    }
    if (!EX && sg == 0) {
      bd = wave_sum(bd);
      if (lane == 0) *(float*)(smem + bd_off + (uint32_t)wave * 4u) = bd;
    }
  }
  // RG: the second queue stage.  Slot S keeps the element word of the block whose residual entry is on its way (ew) and that
  // entry (rgq); `gather` moves the slot's freshly arrived (element word, residual index) there and asks for the entry.  Primed
  // here: blocks 0 .. Q - 1 go to the second stage, blocks Q .. 2 Q - 1 are requested behind them.
  [[maybe_unused]] uint32_t ew[RG ? kSLQueue : 1];
  [[maybe_unused]] u32x4 rgq[RG ? kSLQueue : 1][W4];
  [[maybe_unused]] auto gather = [&](auto slot_c) __attribute__((always_inline)) {
    constexpr int S = decltype(slot_c)::value;
    ew[S] = eq[S][0];
    const char* const ra = (const char*)as_global(P.rcent) + (size_t)rq[S] * kEntry;
#pragma unroll
    for (int w = 0; w < W4; ++w) {
#if VPTQ_SLICED_RGNT
      rgq[S][w] = __builtin_nontemporal_load((const u32x4*)(ra + 16 * w));
#else
      rgq[S][w] = *(const u32x4*)(ra + 16 * w);
#endif
    }
  };
  if constexpr (RG) {
    sl_for_slots<kSLQueue>([&](auto slot_c) {
      gather(slot_c);
      __builtin_amdgcn_sched_barrier(0);
      issue(slot_c);
      __builtin_amdgcn_sched_barrier(0);
    });
  }
  // the DMA and the staging loads are done before anybody reads LDS (the queue loads stay in flight)
  {   // (as a builtin: the compiler sees it and keeps counting from here)
    constexpr int kN = RG ? kSLQueue * (W4 + 2) : kSLQueue * kLoadsPerStep;
    __builtin_amdgcn_s_waitcnt(0x0F70 | (kN & 15) | ((kN >> 4) << 14));   // vmcnt(kN), nothing else
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
#if VPTQ_SLICED_TRACE
  if (lane == 0) trace[1] = __builtin_amdgcn_s_memrealtime();
#endif
  float bdot = 0.f;
  if (!EX && sg == 0) {   // (fixed order: the 16 waves' parts)
    // read by hand: in front of a compiler-generated LDS read here the compiler drained vmcnt - the element queue of
    // every slice-0 workgroup, one memory latency (found in the ISA; the phase stamps had slice 0 finishing last)
    static_assert(kSLWaves == 16, "sum b x: 16 parts");
    u32x4 q0, q1, q2, q3;
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:32\n\t"
                 "ds_read_b128 %3, %4 offset:48\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3) : "v"(bd_off) : "memory");
#pragma unroll
    for (int i = 0; i < 4; ++i) bdot += __uint_as_float(q0[i]);
#pragma unroll
    for (int i = 0; i < 4; ++i) bdot += __uint_as_float(q1[i]);
#pragma unroll
    for (int i = 0; i < 4; ++i) bdot += __uint_as_float(q2[i]);
#pragma unroll
    for (int i = 0; i < 4; ++i) bdot += __uint_as_float(q3[i]);
  }
#if VPTQ_SLICED_PRIO == 1
  __builtin_amdgcn_s_setprio(3);   // (A/B, round 5) constants only: the instruction takes an immediate
  if (wave < 12) __builtin_amdgcn_s_setprio(2);
  if (wave < 8) __builtin_amdgcn_s_setprio(1);
  if (wave < 4) __builtin_amdgcn_s_setprio(0);
#elif VPTQ_SLICED_PRIO == 2
  __builtin_amdgcn_s_setprio(0);
  if (wave < 12) __builtin_amdgcn_s_setprio(1);
  if (wave < 8) __builtin_amdgcn_s_setprio(2);
  if (wave < 4) __builtin_amdgcn_s_setprio(3);
#endif
  float acc[TOK][V];
#pragma unroll
  for (int t = 0; t < TOK; ++t)
#pragma unroll
    for (int i = 0; i < V; ++i) acc[t][i] = 0.f;
  int row_i = 0;
  // ---- the slices of a row meet in the output's accumulator word (caller's workspace, zero between launches): the lane
  // whose add brings the arrivals to NSLT has old + its own = the sum of all slices (+ sum b x, which rides with slice 0),
  // exact in fixed point and therefore the same whoever comes last; it rounds ONCE, adds the output bias, stores y and
  // puts the word back to zero.  One returning atomic is the whole hand-over: nobody waits for anybody, nothing is read back.
  int pend_o = -1;
  unsigned long long pend_old[TOK][V / 4], pend_mine[TOK][V / 4];
  auto finish_rows = [&]() __attribute__((always_inline)) {
    if (pend_o >= 0) {
#pragma unroll
      for (int t = 0; t < TOK; ++t) {
#pragma unroll
        for (int i = 0; i < V / 4; ++i) {
          if ((pend_old[t][i] & 127ull) == (unsigned long long)(GP.arrivals - 1)) {
            const int o = pend_o + i;
            float r = sl_from_fixed<sl_frac<DT>()>(pend_old[t][i] + pend_mine[t][i]);
            if (o < P.O) {
              if (P.corr) r += as_global(P.corr)[o];
              if (P.bias) r += DT::to_float(as_global(P.bias)[o]);
              const size_t yo = (TOK > 1 ? (size_t)t * P.y_stride : 0) + (size_t)o;
              if (P.out_f32) ((float*)as_global(P.y))[yo] = r;
              else ((uint16_t*)as_global(P.y))[yo] = DT::from_float(r);
            }
            __hip_atomic_store((unsigned long long*)as_global(P.partial) + (TOK > 1 ? (size_t)t * P.acc_stride : 0) + o, 0ull, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
          }
        }
      }
      pend_o = -1;
    }
  };
  // rows without elements in this slice store zeros
  auto store_row = [&]() __attribute__((always_inline)) {
    // sum over the 64 lanes: swap-and-add halves the values carried (gemv_k256c.hip:finish), then DPP
    float v[TOK][V];
#pragma unroll
    for (int t = 0; t < TOK; ++t) {
#pragma unroll
      for (int i = 0; i < V; ++i) v[t][i] = acc[t][i];
#pragma unroll
      for (int i = 0; i < V / 2; ++i) {
        auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[t][i]), __float_as_uint(v[t][i + V / 2]), false, false);
        v[t][i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
      }
#pragma unroll
      for (int i = 0; i < V / 4; ++i) {
        auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[t][i]), __float_as_uint(v[t][i + V / 4]), false, false);
        v[t][i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
      }
#pragma unroll
      for (int i = 0; i < V / 4; ++i) v[t][i] = row16_allsum(v[t][i]);
    }
    // lane l (any of its row of 16) holds outputs (V / 2) bit5 + (V / 4) bit4 + {0 .. V / 4 - 1}
    // row i of the wave is handed over by lanes (i & 15) + {0, 16, 32, 48}: up to 16 rows' returned words wait in the
    // registers of different lanes until the stream is through (no wait inside the loop)
    if ((row_i & 15) == 0 && row_i > 0) finish_rows();   // (more than 16 rows per wave: the lanes come round again)
    if ((lane & 15) == (row_i & 15)) {
      pend_o = (row0 + row_i) * V + ((lane >> 5) & 1) * (V / 2) + ((lane >> 4) & 1) * (V / 4);
#pragma unroll
      for (int t = 0; t < TOK; ++t) {
        unsigned long long* const ap = (unsigned long long*)as_global(P.partial) + (TOK > 1 ? (size_t)t * P.acc_stride : 0) + pend_o;
#pragma unroll
        for (int i = 0; i < V / 4; ++i) {
          pend_mine[t][i] = sl_to_fixed<sl_frac<DT>()>(v[t][i] + bdot);
          pend_old[t][i] = __hip_atomic_fetch_add(ap + i, pend_mine[t][i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
#pragma unroll
    for (int t = 0; t < TOK; ++t)
#pragma unroll
      for (int i = 0; i < V; ++i) acc[t][i] = 0.f;
  };
  int left = 0x7fffffff;
  bool done = false;
  if (total == 0) {   // none of this wave's rows has an element in this slice: zeros
    for (; row_i < n_rows; ++row_i) store_row();
    done = true;
  } else {
    left = __builtin_amdgcn_readlane(my_blocks, 0);
    while (left == 0) {   // (leading rows with no element in this slice)
      store_row();
      ++row_i;
      left = __builtin_amdgcn_readlane(my_blocks, row_i);
    }
  }
  // ---- a block's work in two halves: FETCH (its element word -> LDS addresses -> the gathers go out) and MATH (on what a fetch
  // brought).  VPTQ_SLICED_PIPE (round 5): block k's gathers are issued BEFORE block k - 1's arithmetic, into the other of two
  // register sets - a wave then covers its own LDS latency instead of leaving that to the 3 other waves of its SIMD (the
  // phase stamps showed the SIMD's issue bandwidth half idle: 100 cycles per step for 52 cycles of vector work).
  typedef __attribute__((address_space(3))) uint16_t lds_h_t;
  typedef __attribute__((address_space(3))) uint32_t lds_w_t;
  // (measured, profiles/r05/sliced_pipe_ab.txt: -1 ... -3 % without the 256-entry residual table, + 4 % with it - its third
  // gather per block and 8 more live registers: those instantiations keep gather, wait, compute)
  constexpr bool kPipe = VPTQ_SLICED_PIPE != 0 && !RES && !RG;
  constexpr int kBufs = kPipe ? 2 : 1;
  u32x4 g_ent[kBufs][W4];
  u32x4 g_rent[kBufs];
  uint32_t g_x[kBufs], g_sb[kBufs];
  [[maybe_unused]] uint32_t g_x2[kBufs];   // (TOK = 3) token 2
  auto fetch = [&](auto slot_c, auto buf_c) __attribute__((always_inline)) {
    constexpr int S = decltype(slot_c)::value, Bf = decltype(buf_c)::value;
    const uint32_t e = RG ? ew[S] : eq[S][0];
    if constexpr ((VPTQ_SLICED_ABLATE & 1) != 0) { g_x[Bf] = e; return; }
    const uint32_t ea = (e >> 16) * kEntry;
    // the element's staged column (WPT: inside this workgroup's windows, else - other windows' elements, padding - the zero column)
    uint32_t col = e & 0xffffu;
    if constexpr (WPT) {
      col -= (uint32_t)c0;
      col = col < (uint32_t)G ? col : (uint32_t)G;
    }
#pragma unroll
    for (int w = 0; w < W4; ++w) g_ent[Bf][w] = lds_load16(ea + 16u * (uint32_t)w);
    if constexpr (TOK == 1) g_x[Bf] = *(const lds_h_t*)(uintptr_t)(kSLXOff + (col << 1));
    else g_x[Bf] = *(const lds_w_t*)(uintptr_t)(kSLXOff + (col << 2));   // tokens 0 | 1
    if constexpr (TOK == 3) g_x2[Bf] = *(const lds_h_t*)(uintptr_t)(x2_off + (col << 1));
    if constexpr (RES) g_rent[Bf] = lds_load16(res_off + (rq[S] << 4));
    if constexpr (EX) g_sb[Bf] = *(const lds_w_t*)(uintptr_t)(sb_off + (col << 2));
  };
  auto math = [&](auto buf_c, auto slot_c) __attribute__((always_inline)) {
    constexpr int Bf = decltype(buf_c)::value;
    [[maybe_unused]] constexpr int S = decltype(slot_c)::value;   // (RG: the residual entry sits in the slot's second stage)
    if constexpr ((VPTQ_SLICED_ABLATE & 1) != 0) { acc[0][0] += __uint_as_float(g_x[Bf]); return; }
    // EX: the weight as the reference rounds it - u = f16(c + r), t = f16(u * s), w = f16(t + b): three packed instructions per
    // pair of outputs, scale and bias broadcast out of the column's word by op_sel - then w x in fp32
    auto weight = [&](uint32_t ew, uint32_t rw, uint32_t sbw) __attribute__((always_inline)) -> uint32_t {
      if constexpr (RES || RG) ew = DT::add2_g(ew, rw);
      if constexpr (EX) {
        ew = DT::mul2_bcast_g(ew, sbw, 0);
        ew = DT::add2_bcast_g(ew, sbw, 1);
      }
      return ew;
    };
    if constexpr (std::is_same<DT, F16>::value) {
      // v_fma_mix_f32: fp16 x fp16 + fp32 -> fp32 in one instruction (exact product, one rounding: what
      // fmaf of the converted values gives), halves picked by op_sel
      const uint32_t xw = g_x[Bf];
#pragma unroll
      for (int i = 0; i < V / 2; ++i) {
        float lo = acc[0][2 * i], hi = acc[0][2 * i + 1];   // (an asm operand cannot name a captured array element)
        // 256-entry residual table: f16(c + r) first - the reference's own first rounding (vptq/ops/quant_gemm.py:121) -
        // as ONE packed add per pair of outputs instead of a second pair of multiply-adds (round 5: the phase stamps
        // showed this format's stream bound by vector issue, 25 instructions per block)
        const uint32_t ew = weight(g_ent[Bf][i / 4][i % 4], RES ? g_rent[Bf][i % 4] : (RG ? rgq[RG ? S : 0][i / 4][i % 4] : 0u), EX ? g_sb[Bf] : 0u);
        asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,1,0]" : "+v"(lo) : "v"(ew), "v"(xw));
        asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,0]" : "+v"(hi) : "v"(ew), "v"(xw));
        acc[0][2 * i] = lo; acc[0][2 * i + 1] = hi;
        if constexpr (TOK >= 2) {   // token 1: the high half of the same word
          float lo1 = acc[1][2 * i], hi1 = acc[1][2 * i + 1];
          asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,0]" : "+v"(lo1) : "v"(ew), "v"(xw));
          asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,1,0]" : "+v"(hi1) : "v"(ew), "v"(xw));
          acc[1][2 * i] = lo1; acc[1][2 * i + 1] = hi1;
        }
        if constexpr (TOK == 3) {
          const uint32_t xw2 = g_x2[Bf];
          float lo2 = acc[2][2 * i], hi2 = acc[2][2 * i + 1];
          asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,1,0]" : "+v"(lo2) : "v"(ew), "v"(xw2));
          asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,0]" : "+v"(hi2) : "v"(ew), "v"(xw2));
          acc[2][2 * i] = lo2; acc[2][2 * i + 1] = hi2;
        }
      }
    } else {
      // bf16: the widened arithmetic straight from the packed pairs, four pairs per scheduled block (common.h: BF16::add4 /
      // scale_bias4 / fma4, round 6): 8 dot instructions + 4 conversions per rounding stage, 8 dot instructions per token's products
      const uint32_t xw = g_x[Bf];   // token 0 (| token 1 << 16)
#pragma unroll
      for (int g4 = 0; g4 < V / 8; ++g4) {
        uint32_t w[4] = {g_ent[Bf][g4][0], g_ent[Bf][g4][1], g_ent[Bf][g4][2], g_ent[Bf][g4][3]};
        [[maybe_unused]] uint32_t r[4] = {0u, 0u, 0u, 0u};
        if constexpr (RES) { r[0] = g_rent[Bf][0]; r[1] = g_rent[Bf][1]; r[2] = g_rent[Bf][2]; r[3] = g_rent[Bf][3]; }
        if constexpr (RG) { r[0] = rgq[RG ? S : 0][g4][0]; r[1] = rgq[RG ? S : 0][g4][1]; r[2] = rgq[RG ? S : 0][g4][2]; r[3] = rgq[RG ? S : 0][g4][3]; }
        if constexpr (EX) {
          if constexpr (RES || RG) BF16::add4(w, r);
          BF16::scale_bias4(w, g_sb[Bf], 0, g_sb[Bf], 1);
        }
        BF16::fma4(&acc[0][8 * g4], w, xw, 0);
        if constexpr (TOK >= 2) BF16::fma4(&acc[TOK >= 2 ? 1 : 0][8 * g4], w, xw, 1);
        if constexpr (TOK == 3) BF16::fma4(&acc[TOK == 3 ? 2 : 0][8 * g4], w, g_x2[Bf], 0);
        // (folded bf16: c x + r x, two sets of multiply-adds - a widened add would cost more than it saves)
        if constexpr (RES && !EX) BF16::fma4(&acc[0][8 * g4], r, xw, 0);
      }
    }
  };
  // the end of a row: `left` counts the blocks of the current row whose arithmetic is still to come
  auto row_step = [&]() __attribute__((always_inline)) {
    if (--left == 0) {
      if (!done) store_row();
      ++row_i;
      // next row with elements; after the last one the remaining steps of this loop iteration only keep the loads
      // counted (ONE loop exit, at the end: gemv_k256c.hip)
      left = 0;
      while (row_i < n_rows && left == 0) {
        left = __builtin_amdgcn_readlane(my_blocks, row_i);
        if (left == 0) { store_row(); ++row_i; }
      }
      if (row_i >= n_rows) { done = true; left = 0x7fffffff; }
    }
  };
  static_assert(kSLQueue % 2 == 0 && EPL == 1, "two register sets alternate over an even number of slots");
  // pipelined: block 0's gathers go out in front of the loop; the loop's step for slot S fetches THAT slot's block and then
  // does the arithmetic of the block before it (slot S - 1): slots are walked 1, 2, ... Q - 1, 0
  if constexpr (kPipe) {
    if (!done) {
      fetch(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
      __builtin_amdgcn_sched_barrier(0);
      issue(std::integral_constant<int, 0>{});
    }
  }
  auto step = [&](auto slot_c) __attribute__((always_inline)) {
    constexpr int S = kPipe ? (decltype(slot_c)::value + 1) % kSLQueue : decltype(slot_c)::value;
    __builtin_amdgcn_sched_barrier(0);
    // (past the end of the wave's stream the steps of the last round only keep the loads counted: a stream is
    // ~34 blocks long, so up to depth - 1 idle steps were 15 % of the launch while they still gathered and added)
    if (!done) {
      if constexpr (kPipe) {
        fetch(std::integral_constant<int, S>{}, std::integral_constant<int, S & 1>{});
        __builtin_amdgcn_sched_barrier(0);
        math(std::integral_constant<int, (S & 1) ^ 1>{}, std::integral_constant<int, 0>{});
      } else {
        fetch(std::integral_constant<int, S>{}, std::integral_constant<int, 0>{});
        math(std::integral_constant<int, 0>{}, std::integral_constant<int, S>{});
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (RG) {   // the slot's (element word, residual index) - requested a round ago - move to the second stage
      gather(std::integral_constant<int, S>{});
      __builtin_amdgcn_sched_barrier(0);
    }
    issue(std::integral_constant<int, S>{});
    __builtin_amdgcn_sched_barrier(0);
    row_step();
  };
  while (!done) {   // (waves without elements skip it; ONE exit for the others)
    sl_for_slots<kSLQueue>(step);
  }

  // ---- the rows whose words are still on their way: the last arriver of every output rounds and stores it
#if VPTQ_SLICED_TRACE
  if (lane == 0) trace[2] = __builtin_amdgcn_s_memrealtime();
#endif
  finish_rows();
#if VPTQ_SLICED_TRACE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane == 0) trace[3] = __builtin_amdgcn_s_memrealtime();
#endif
}

// ---- launchers (both parts of the build: the file is compiled twice, VPTQ_SL_PART = 1: the one-token instantiations + the host
// side, 2: the 2 / 3-token instantiations of the reference's roundings)
template <typename DT, int NSL, bool RES, int V, bool TWO, bool EX = false, bool RG = false, int TOK = 1, bool WPT = false>
static hipError_t launch_sl(const SlicedGroupParams& P, uint32_t lds, hipStream_t st) {
  auto kern = gemv_sliced_kernel<DT, NSL, RES, V, TWO, EX, RG, TOK, WPT>;
  static std::atomic<bool> attr_set[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (!attr_set[dev]) {
    const hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSLLdsLimit);
    if (e != hipSuccess) return e;
    attr_set[dev] = true;
  }
  hipLaunchKernelGGL(kern, dim3(P.start[P.n]), dim3(kSLThreads), lds, st, P);
  return hipGetLastError();
}
hipError_t launch_sl_tokens(int dtype, const SlicedGroupParams& P, int v, int nsl, bool res, bool rg, int tokens, uint32_t lds, hipStream_t st);
hipError_t launch_sl_tokens_wpt(int dtype, const SlicedGroupParams& P, int nsl, bool res, int tokens, uint32_t lds, hipStream_t st);
#if !defined(VPTQ_SL_PART) || VPTQ_SL_PART == 3
// (window parts: v = 8, one table)
template <typename DT, int TOK>
static hipError_t launch_sl_tok_wpt(const SlicedGroupParams& P, int nsl, bool res, uint32_t lds, hipStream_t st) {
  if (nsl == 8) return res ? launch_sl<DT, 8, true, 8, false, true, false, TOK, true>(P, lds, st) : launch_sl<DT, 8, false, 8, false, true, false, TOK, true>(P, lds, st);
  return res ? launch_sl<DT, 16, true, 8, false, true, false, TOK, true>(P, lds, st) : launch_sl<DT, 16, false, 8, false, true, false, TOK, true>(P, lds, st);
}
hipError_t launch_sl_tokens_wpt(int dtype, const SlicedGroupParams& P, int nsl, bool res, int tokens, uint32_t lds, hipStream_t st) {
  if (tokens == 2) return dtype == VPTQ_DTYPE_F16 ? launch_sl_tok_wpt<F16, 2>(P, nsl, res, lds, st) : launch_sl_tok_wpt<BF16, 2>(P, nsl, res, lds, st);
  if (tokens == 3) return dtype == VPTQ_DTYPE_F16 ? launch_sl_tok_wpt<F16, 3>(P, nsl, res, lds, st) : launch_sl_tok_wpt<BF16, 3>(P, nsl, res, lds, st);
  return hipErrorInvalidValue;
}
#endif
#if !defined(VPTQ_SL_PART) || VPTQ_SL_PART == 2
template <typename DT, int TOK>
static hipError_t launch_sl_tok(const SlicedGroupParams& P, int v, int nsl, bool res, bool rg, uint32_t lds, hipStream_t st) {
  if (rg) {   // (v = 8 only: the gathered residual entries of v = 16 on top of two tokens' sums spill)
    if (v != 8) return hipErrorInvalidValue;
    return nsl == 8 ? launch_sl<DT, 8, false, 8, false, true, true, TOK>(P, lds, st) : launch_sl<DT, 16, false, 8, false, true, true, TOK>(P, lds, st);
  }
  if (v == 16) {   // (3 tokens of 16 outputs: 48 sums + 12 returned words per lane spill - 2 tokens only)
    if constexpr (TOK == 2) return nsl == 16 ? launch_sl<DT, 16, false, 16, false, true, false, TOK>(P, lds, st) : launch_sl<DT, 32, false, 16, false, true, false, TOK>(P, lds, st);
    else return hipErrorInvalidValue;
  }
  if (nsl == 8) return res ? launch_sl<DT, 8, true, 8, false, true, false, TOK>(P, lds, st) : launch_sl<DT, 8, false, 8, false, true, false, TOK>(P, lds, st);
  return res ? launch_sl<DT, 16, true, 8, false, true, false, TOK>(P, lds, st) : launch_sl<DT, 16, false, 8, false, true, false, TOK>(P, lds, st);
}
hipError_t launch_sl_tokens(int dtype, const SlicedGroupParams& P, int v, int nsl, bool res, bool rg, int tokens, uint32_t lds, hipStream_t st) {
  if (tokens == 2) return dtype == VPTQ_DTYPE_F16 ? launch_sl_tok<F16, 2>(P, v, nsl, res, rg, lds, st) : launch_sl_tok<BF16, 2>(P, v, nsl, res, rg, lds, st);
  if (tokens == 3) return dtype == VPTQ_DTYPE_F16 ? launch_sl_tok<F16, 3>(P, v, nsl, res, rg, lds, st) : launch_sl_tok<BF16, 3>(P, v, nsl, res, rg, lds, st);
  return hipErrorInvalidValue;
}
#endif

#if !defined(VPTQ_SL_PART) || VPTQ_SL_PART == 1
// ---- host side -------------------------------------------------------------------
// Large-codebook layers: v = 8 or 16, 16384 ... 65536 main centroids, one codebook group, no outlier columns, norm on.
// Residual codebook: none; v = 8 with 256 entries (one launch, the table beside the slice, a byte per element); any other
// size (2 ... 65536 entries) as a SECOND TABLE with a layout of its own in the same launch.
static bool sl_pow2(int k) { return k > 0 && (k & (k - 1)) == 0; }
bool sl_res256(const VptqLayerDesc& d) { return d.vector_len == 8 && d.num_res_centroids == 256; }
bool sl_two(const VptqLayerDesc& d) { return d.num_res_centroids > 0 && !sl_res256(d); }
int gemv_sliced_tables(const VptqLayerDesc& d) { return sl_two(d) ? 2 : 1; }
static bool sl_shape_ok(const VptqLayerDesc& d, bool exact);
bool gemv_sliced_eligible(const VptqLayerDesc& d, bool exact) {
  const int T = d.index_bits + d.res_bits;
  // the reference's roundings: bias in column order (c and r meet in one lane: a residual codebook other than v8's 256-entry
  // one is gathered from L2 by a side stream of 16-bit indices - ONE layout, bucketed by the main index)
  if (exact && d.perm != nullptr && (d.bias_permuted == nullptr || (((uintptr_t)d.bias_permuted) & 15) != 0)) return false;
  return sl_shape_ok(d, exact) && (d.vector_len == 8 || d.vector_len == 16) && d.num_codebooks == 1 && d.outlier_size == 0 &&
         d.num_centroids >= 16384 && d.num_centroids <= 65536 && sl_pow2(d.num_centroids) && (1 << d.index_bits) == d.num_centroids &&
         (d.num_res_centroids == 0 || (d.num_res_centroids >= 2 && d.num_res_centroids <= 65536 && sl_pow2(d.num_res_centroids) &&
                                       (1 << d.res_bits) == d.num_res_centroids)) &&
         d.weight_scale != nullptr && d.weight_bias != nullptr &&
         (d.perm == nullptr || d.scale_permuted != nullptr) && (d.group_size % 8) == 0 && d.group_size == d.in_features &&
         d.group_size <= kSLMaxG16 && T <= 32 && (long long)d.row_words * 32 >= (long long)d.group_size * T &&
         (((uintptr_t)d.centroids | (uintptr_t)d.res_centroids | (uintptr_t)d.weight_scale | (uintptr_t)d.weight_bias |
           (uintptr_t)d.perm | (uintptr_t)d.scale_permuted) & 15) == 0;
}

// bytes of LDS behind the table: f16(s x) of every column (+ 64 padding columns) + 16 floats; the reference's roundings
// stage x and a word {scale, bias} per column instead: 6 bytes per column; + the 4 KiB residual table of the 256-entry path
static int sl_window_cols(const VptqLayerDesc& d) { return (d.group_size + kSLWindows * 8 - 1) / (kSLWindows * 8) * 8; }   // (the layout's window width)
static uint32_t sl_operand_bytes(const VptqLayerDesc& d, bool exact, int tokens = 1, int columns = 0) {
  return (uint32_t)((columns ? columns : d.group_size) + 64) * (exact ? 4u + 2u * (uint32_t)tokens : 2u) + 64u + (sl_res256(d) ? 4096u : 0u);
}
// slices a layout of this layer must have: the slice (table entries / slices, 2 v bytes each) + the staged operands must fit
// the 160 KiB of LDS.  Folded arithmetic: v = 8: 8 slices up to 14336 columns (14080 with the 256-entry table), else 16;
// v = 16: 16 slices up to 14336 columns, else 32.  Reference roundings (6 bytes per column): v = 8: 8 slices up to 5376 columns
// (4704), 16 up to 16288 (15616); v = 16: 16 / 32 at the same widths; wider layers: 0 (not served: gemv_gather)
int gemv_sliced_slices(const VptqLayerDesc& d, bool exact) {
  static std::atomic<int> force16{-1};   // VPTQ_SLICED_SLICES=16: the larger slice count for every layer (A/B)
  if (force16 < 0) { const char* e = vptq::tune_env("VPTQ_SLICED_SLICES"); force16 = (e && atoi(e) == 16) ? 1 : 0; }
  const int small = d.vector_len == 16 ? 16 : 8;
  if (!exact) {
    if (force16 == 1) return 2 * small;
    return d.group_size <= (sl_res256(d) ? kSLMaxG8Res : kSLMaxG8) ? small : 2 * small;
  }
  if (d.num_centroids <= 0 || d.vector_len <= 0) return 0;
  // VPTQ_SLICED_SLICES=room2 (process-wide opt-in, v = 8): the smaller count only where the slice leaves room for TWO tokens'
  // operands, so that 2 / 3 tokens take one pass of the one-token kernel (TOK) on 4096-column layers too - they miss that by 0.5 KiB
  // with 128 KiB slices.  Measured on the Llama-3-8B shapes in v8-k65536-256 (profiles/r05/sliced_exact_slices_16_for_narrow_layers.txt):
  // 2 sequences 198.6 -> 210.7 tokens/s, 3: 270.6 -> 277.3, ONE: 127.1 -> 125.4 (the layers' own time +5 %) - one token is the
  // default's business, so the default stays "8 slices wherever one token fits".
  static std::atomic<int> room2{-1};
  if (room2 < 0) { const char* e = vptq::tune_env("VPTQ_SLICED_SLICES"); room2 = (e && strcmp(e, "room2") == 0) ? 1 : 0; }
  for (int nsl = (force16 == 1 ? 2 * small : small); nsl <= 2 * small; nsl *= 2) {
    const uint32_t tab = (uint32_t)(d.num_centroids / nsl) * (uint32_t)d.vector_len * 2u;
    if ((tab + 15u) / 16u * 16u + sl_operand_bytes(d, true, (nsl == small && room2 == 1 && d.vector_len == 8) ? 2 : 1) <= kSLLdsLimit) return nsl;
  }
  return 0;
}
// bytes a workgroup of a table with k entries holds: its slice, or (whole != 0) the whole table
uint32_t sl_tab_bytes(const VptqLayerDesc& d, int k, int whole, bool exact) {
  const int nsl = gemv_sliced_slices(d, exact);
  return (uint32_t)(whole || nsl == 0 ? k : k / nsl) * (uint32_t)d.vector_len * 2u;
}
// Does every workgroup of table t (0 main, 1 residual-as-second-table) hold the WHOLE table (element words then carry the
// full index and a row's elements are split into column ranges)?  Only a second table whose slice would be under 16 KiB -
// copied in 1 KiB pieces by 16 waves, a smaller slice is mostly partial pieces and its lists are short - and only while the
// whole table still fits beside the staged activations.
int gemv_sliced_whole_table(const VptqLayerDesc& d, int t) {
  if (t != 1 || !sl_two(d)) return 0;
  const uint32_t slice = sl_tab_bytes(d, d.num_res_centroids, 0), whole = sl_tab_bytes(d, d.num_res_centroids, 1);
  const uint32_t main_slice = sl_tab_bytes(d, d.num_centroids, 0);
  const uint32_t x_bytes = (uint32_t)(d.group_size + 64) * 2u + 64u;
  if (d.num_res_centroids < gemv_sliced_slices(d)) return 1;   // (fewer entries than slices: no other way)
  return slice < 16384u && ((whole > main_slice ? whole : main_slice) + 15u) / 16u * 16u + x_bytes <= kSLLdsLimit ? 1 : 0;
}

// the tables' parts + the staged operands fit the LDS
static bool sl_shape_ok(const VptqLayerDesc& d, bool exact) {
  if (!(d.vector_len == 8 || d.vector_len == 16) || d.group_size <= 0 || d.group_size > kSLMaxG16 || d.num_centroids < 16384) return false;
  if (exact) return gemv_sliced_slices(d, true) != 0;
  uint32_t tab = sl_tab_bytes(d, d.num_centroids, 0);
  if (sl_two(d)) {
    const uint32_t t1 = sl_tab_bytes(d, d.num_res_centroids, gemv_sliced_whole_table(d, 1));
    tab = t1 > tab ? t1 : tab;
  }
  return (tab + 15u) / 16u * 16u + sl_operand_bytes(d, false) <= kSLLdsLimit;
}

// one 64-bit accumulator word per output (N x v of them), which must be ZERO before the first launch; every launch leaves
// them zero.  (Rounds 3-4 kept [tables x slices][N x v] float partial sums + arrival counters here; the size the ABI asks for
// is still that one - callers' buffers stay valid, tools/sliced_trace.py stamps into the rest.)
size_t gemv_sliced_workspace_bytes(const VptqLayerDesc& d) {
  const size_t parts = (size_t)gemv_sliced_slices(d) * (sl_two(d) ? 2 : 1);
  const size_t partial = (parts * d.num_indices * d.vector_len * sizeof(float) + 255) / 256 * 256;
  const size_t counters = (((size_t)(d.num_indices + kSLWaves - 1) / kSLWaves) * sizeof(uint32_t) + 255) / 256 * 256;
  return partial + counters;
}

template <typename DT>
static hipError_t launch_sl_dt(const SlicedGroupParams& P, int v, int nsl, bool res, bool two, bool exact, uint32_t lds, hipStream_t st) {
  if (exact) {   // the reference's roundings: one layout; another residual codebook than the 256-entry one comes from L2 (RG)
    if (two) {
      if (v == 16) return nsl == 16 ? launch_sl<DT, 16, false, 16, false, true, true>(P, lds, st) : launch_sl<DT, 32, false, 16, false, true, true>(P, lds, st);
      return nsl == 8 ? launch_sl<DT, 8, false, 8, false, true, true>(P, lds, st) : launch_sl<DT, 16, false, 8, false, true, true>(P, lds, st);
    }
    if (v == 16) return nsl == 16 ? launch_sl<DT, 16, false, 16, false, true>(P, lds, st) : launch_sl<DT, 32, false, 16, false, true>(P, lds, st);
    if (nsl == 8) return res ? launch_sl<DT, 8, true, 8, false, true>(P, lds, st) : launch_sl<DT, 8, false, 8, false, true>(P, lds, st);
    return res ? launch_sl<DT, 16, true, 8, false, true>(P, lds, st) : launch_sl<DT, 16, false, 8, false, true>(P, lds, st);
  }
  if (v == 16) {
    if (two) return nsl == 16 ? launch_sl<DT, 16, false, 16, true>(P, lds, st) : launch_sl<DT, 32, false, 16, true>(P, lds, st);
    return nsl == 16 ? launch_sl<DT, 16, false, 16, false>(P, lds, st) : launch_sl<DT, 32, false, 16, false>(P, lds, st);
  }
  if (two) return nsl == 8 ? launch_sl<DT, 8, false, 8, true>(P, lds, st) : launch_sl<DT, 16, false, 8, true>(P, lds, st);
  if (nsl == 8) return res ? launch_sl<DT, 8, true, 8, false>(P, lds, st) : launch_sl<DT, 8, false, 8, false>(P, lds, st);
  return res ? launch_sl<DT, 16, true, 8, false>(P, lds, st) : launch_sl<DT, 16, false, 8, false>(P, lds, st);
}

bool sl_layout_ok(const VptqLayerDesc& d, const VptqSlicedLayout& L, int nsl, bool res, int k) {
  // (a sliced table needs at least one entry per slice; whole = every workgroup of the table holds all of it)
  return (L.n_slices != 0 ? L.n_slices : 8) == nsl && (L.elems_per_lane == 0 || L.elems_per_lane == 1) && (!res || L.res) &&
         L.rows_per_wave >= 1 && L.rows_per_wave <= kSLMaxRowsPerWave && L.elems && L.blocks && L.first &&
         (((uintptr_t)L.elems) & 3) == 0 && (L.whole_table == 0 || L.whole_table == 1) && (L.whole_table || k >= nsl);
}

// L: one layout (residual none / the 256-entry path of v = 8) or TWO consecutive ones (any other residual codebook: [0]
// bucketed by the main index, [1] by the residual index).  (c + r) f16(s x) = c f16(s x) + r f16(s x): the residual table's
// (slice, row block) workgroups run beside the main table's in the SAME launch and meet them in the output's accumulator
// word - two launches, one per table, cost a second boundary, a second epilogue and half the workgroups in flight (8192^2: 27.2 us
// against 21.2; 4096^2: 17.8 against 12.0)
static hipError_t sl_fill(const VptqLayerDesc& d, const VptqSlicedLayout* L, const void* x, void* y, int flags, void* ws,
                          SlicedParams& P, uint32_t& lds, int tokens = 1) {
  const bool exact = (flags & VPTQ_GEMV_EXACT) != 0;
  if (tokens != 1 && !gemv_sliced_exact_tokens_ok(d, tokens)) return hipErrorInvalidValue;
  const bool res = sl_res256(d), rg = exact && sl_two(d), two = sl_two(d) && !exact;
  const int nsl = gemv_sliced_slices(d, exact);
  if (nsl == 0 || !sl_layout_ok(d, L[0], nsl, res || rg, d.num_centroids) || L[0].whole_table != 0 ||
      (rg && (((uintptr_t)L[0].res) & 1) != 0) ||
      (two && (!sl_layout_ok(d, L[1], nsl, false, d.num_res_centroids) || L[1].rows_per_wave != L[0].rows_per_wave ||
               L[1].whole_table != gemv_sliced_whole_table(d, 1))) ||
      !ws || (((uintptr_t)x) & 15) != 0)
    return hipErrorInvalidValue;
  P = SlicedParams{};
  P.elems = (const uint32_t*)L[0].elems;
  P.res = (res || rg) ? (const uint8_t*)L[0].res : nullptr;   // (rg: uint16 per element)
  P.rcent = (res || rg) ? (const uint32_t*)d.res_centroids : nullptr;
  P.blocks = (const int32_t*)L[0].blocks;
  P.first = (const int32_t*)L[0].first;
  P.cent = (const uint32_t*)d.centroids;
  P.tab0 = sl_tab_bytes(d, d.num_centroids, L[0].whole_table, exact);
  P.stride0 = L[0].whole_table ? 0u : P.tab0;
  if (two) {
    P.elems2 = (const uint32_t*)L[1].elems;
    P.blocks2 = (const int32_t*)L[1].blocks;
    P.first2 = (const int32_t*)L[1].first;
    P.cent2 = (const uint32_t*)d.res_centroids;
    P.tab1 = sl_tab_bytes(d, d.num_res_centroids, L[1].whole_table);
    P.stride1 = L[1].whole_table ? 0u : P.tab1;
  }
  P.x_off = ((P.tab0 > P.tab1 ? P.tab0 : P.tab1) + 15u) & ~15u;
  P.x = (const uint16_t*)x;
  P.scale = (const uint16_t*)(d.perm ? d.scale_permuted : d.weight_scale);
  P.wbias = (const uint16_t*)d.weight_bias;
  P.cbias = (const uint16_t*)(d.perm ? d.bias_permuted : d.weight_bias);
  P.perm = (const uint16_t*)d.perm;
  P.bias = (const uint16_t*)d.bias;
  P.partial = (float*)ws;
  P.y = y;
  P.N = d.num_indices; P.G = d.group_size; P.O = d.out_features;
  P.rows_per_wave = L[0].rows_per_wave;
  const int rows_per_wg = kSLWaves * L[0].rows_per_wave;
  P.n_rowblocks = (d.num_indices + rows_per_wg - 1) / rows_per_wg;
  P.out_f32 = (flags & VPTQ_GEMV_OUT_F32) ? 1 : 0;
  P.x_stride = d.in_features; P.y_stride = d.out_features; P.acc_stride = d.num_indices * d.vector_len;
  P.wparts = 1;
  if (tokens != 1) {
    P.wparts = gemv_sliced_exact_tokens_parts(d, tokens);
    if (P.wparts > 1) {
      if (!L[0].wstart) return hipErrorInvalidValue;
      P.wstart = (const int32_t*)L[0].wstart;
      P.wcols = sl_window_cols(d);
      P.wstage = (kSLWindows / P.wparts) * P.wcols < d.group_size ? (kSLWindows / P.wparts) * P.wcols : d.group_size;
      // (one round of workgroups: the rows per wave grow with the parts)
      const int rpw = L[0].rows_per_wave * P.wparts;
      P.rows_per_wave = rpw > kSLMaxRowsPerWave ? kSLMaxRowsPerWave : rpw;
      const int rows_per_wg2 = kSLWaves * P.rows_per_wave;
      P.n_rowblocks = (d.num_indices + rows_per_wg2 - 1) / rows_per_wg2;
    }
  }
  lds = P.x_off + sl_operand_bytes(d, exact, tokens, P.wparts > 1 ? P.wstage : 0);
  return lds > kSLLdsLimit ? hipErrorInvalidValue : hipSuccess;
}

// n <= kSLMaxGroup layers of ONE format (vector length, slices, residual kind, dtype) and one input width reading the same x:
// layouts = the layers' layout structs one after the other (1 or 2 each); one launch
bool gemv_sliced_groupable(const VptqLayerDesc* d, int n, bool exact) {
  if (n < 1 || n > kSLMaxGroup) return false;
  for (int i = 0; i < n; ++i) {
    if (!gemv_sliced_eligible(d[i], exact)) return false;
    if (d[i].dtype != d[0].dtype || d[i].vector_len != d[0].vector_len || d[i].group_size != d[0].group_size ||
        gemv_sliced_slices(d[i], exact) != gemv_sliced_slices(d[0], exact) || sl_res256(d[i]) != sl_res256(d[0]) || sl_two(d[i]) != sl_two(d[0]))
      return false;
  }
  return true;
}
// 2 / 3 tokens in one pass of the exact kernel (TOK): one-table formats whose slice + (2 tokens + 4) bytes per column fit the LDS.
// Measured (profiles/r05/sliced_exact_tokens_one_pass.txt) where the 16 (v = 16: 32) slice layouts are: wider than ~5000 columns.
// ... in how many WINDOW PARTS (workgroups per (slice, row block), each staging its column windows alone): 1 where all columns
// fit, else - v = 8, one table; the layout must carry `wstart` - 2 or 4; 0 = not served.  VPTQ_SLICED_WINDOW_PARTS=0: never more than 1 (A/B)
int gemv_sliced_exact_tokens_parts(const VptqLayerDesc& d, int tokens) {
  if (tokens < 2 || tokens > (d.vector_len == 16 ? 2 : 3) || !gemv_sliced_eligible(d, true) || (sl_two(d) && d.vector_len != 8)) return 0;
  const int nsl = gemv_sliced_slices(d, true);
  if (nsl == 0) return 0;
  const uint32_t tab = (sl_tab_bytes(d, d.num_centroids, 0, true) + 15u) / 16u * 16u;
  if (tab + sl_operand_bytes(d, true, tokens) <= kSLLdsLimit) return 1;
  static std::atomic<int> on{-1};
  if (on < 0) { const char* e = vptq::tune_env("VPTQ_SLICED_WINDOW_PARTS"); on = (e && e[0] == '0') ? 0 : 1; }
  if (!on || d.vector_len != 8 || sl_two(d)) return 0;
  const int wc = sl_window_cols(d);
  for (int wparts = 2; wparts <= kSLWindows; wparts *= 2) {
    const int wmax = (kSLWindows / wparts) * wc < d.group_size ? (kSLWindows / wparts) * wc : d.group_size;
    if (tab + sl_operand_bytes(d, true, tokens, wmax) <= kSLLdsLimit && nsl * wparts <= 127) return wparts;
  }
  return 0;
}
bool gemv_sliced_exact_tokens_ok(const VptqLayerDesc& d, int tokens) { return gemv_sliced_exact_tokens_parts(d, tokens) >= 1; }
// accumulator words of such a launch: [tokens][N x v], zero before the first launch, left zero by every launch
size_t gemv_sliced_exact_tokens_workspace_bytes(const VptqLayerDesc& d, int tokens) {
  return ((size_t)tokens * d.num_indices * d.vector_len * sizeof(unsigned long long) + 255) / 256 * 256;
}
// VPTQ_GEMV_COLUMN_PARTS: the n "layers" are the column ranges [i G / n, (i + 1) G / n) of ONE layer too wide for the LDS (28672
// columns in the reference's roundings: 6 bytes per column) - descriptors with in_features = group_size = G / n and the column
// order tensors (scale, bias, permutation) advanced to the part's first column, a layout per part built from those columns of
// the index matrix; x is the WHOLE activation (part i reads it from column i G / n on, or through its slice of the permutation),
// y and the accumulator words are shared: an output is complete after n x slices arrivals.
hipError_t launch_gemv_sliced_group(const VptqLayerDesc* d, const VptqSlicedLayout* L, int n, const void* x, void* const* y,
                                    int flags, void* const* ws, hipStream_t st, int tokens, const float* corr) {
  const bool exact = (flags & VPTQ_GEMV_EXACT) != 0;
  const bool parts = (flags & VPTQ_GEMV_COLUMN_PARTS) != 0;
  if (!gemv_sliced_groupable(d, n, exact) || (tokens != 1 && !exact)) return hipErrorInvalidValue;
  if (parts) {
    for (int i = 1; i < n; ++i)
      if (y[i] != y[0] || ws[i] != ws[0] || d[i].out_features != d[0].out_features || d[i].num_indices != d[0].num_indices || d[i].bias != d[0].bias)
        return hipErrorInvalidValue;
    if (!exact) return hipErrorInvalidValue;   // (the folded form stages 32768 columns in one piece: no parts needed)
  }
  SlicedGroupParams GP = {};
  GP.n = n;
  uint32_t lds = 0;
  const int tables = exact ? 1 : gemv_sliced_tables(d[0]);
  const int nsl = gemv_sliced_slices(d[0], exact);
  const int nslt = nsl * tables;
  for (int i = 0; i < n; ++i) {
    uint32_t l = 0;
    // (a part without a permutation reads its own columns of x; with one, its slice of `perm` indexes the whole activation)
    const void* const xi = (parts && d[i].perm == nullptr) ? (const void*)((const uint16_t*)x + (size_t)i * d[i].in_features) : x;
    const hipError_t e = sl_fill(d[i], L + (size_t)i * tables, xi, y[i], flags, ws[i], GP.p[i], l, tokens);
    if (e != hipSuccess) return e;
    if (parts) GP.p[i].x_stride = n * d[i].in_features;   // (tokens of the WHOLE activation: a part's columns lie one row of all parts apart)
    GP.p[i].corr = (n == 1 && tokens == 1 && !exact) ? corr : nullptr;   // (selective roundings: one layer, one token, folded form)
    lds = l > lds ? l : lds;
    GP.start[i + 1] = GP.start[i] + nslt * GP.p[i].wparts * GP.p[i].n_rowblocks;
    if (GP.p[i].wparts != GP.p[0].wparts) return hipErrorInvalidValue;   // (one input width: one answer)
  }
  for (int i = n; i < kSLMaxGroup; ++i) GP.start[i + 1] = GP.start[n];
  GP.arrivals = nslt * (parts ? n : 1) * GP.p[0].wparts;
  if (GP.arrivals > 127) return hipErrorInvalidValue;   // (7 bits of the accumulator word count them)
  if (tokens != 1 && GP.p[0].wparts > 1) return launch_sl_tokens_wpt(d[0].dtype, GP, nsl, sl_res256(d[0]), tokens, lds, st);
  if (tokens != 1) return launch_sl_tokens(d[0].dtype, GP, d[0].vector_len, nsl, sl_res256(d[0]), sl_two(d[0]), tokens, lds, st);
  return d[0].dtype == VPTQ_DTYPE_F16
             ? launch_sl_dt<F16>(GP, d[0].vector_len, nsl, sl_res256(d[0]), sl_two(d[0]), exact, lds, st)
             : launch_sl_dt<BF16>(GP, d[0].vector_len, nsl, sl_res256(d[0]), sl_two(d[0]), exact, lds, st);   // (exact && two = RG)
}
hipError_t launch_gemv_sliced(const VptqLayerDesc& d, const VptqSlicedLayout* L, const void* x, void* y, int flags,
                              void* ws, hipStream_t st, const float* corr) {
  void* const ys[1] = {y};
  void* const wss[1] = {ws};
  return launch_gemv_sliced_group(&d, L, 1, x, ys, flags, wss, st, 1, corr);
}
#endif   // part 1

}  // namespace vptq
