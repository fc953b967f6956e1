// Fused dequant + GEMV for the canonical "2-bit" VPTQ format (v = 8, 256 + 256 centroids),
// persistent MFMA-accumulate variant of gemv_k256.hip.  Same contract, same reference
// (csrc/kernels/quant_gemv.cuh:11-186, csrc/quant_gemv.cu:203-235).
//
// Why a second kernel: gemv_k256.hip is bound by VALU issue (22 instructions per index in
// the exact form, 14 folded; tools/ubench.hip) and, on large layers, by paying its prologue
// (64 KiB image, activations) once per 1-2 vector-rows.  Here
//  * the multiply-accumulate moves to the matrix pipe.  A GEMV is no contraction for MFMA,
//    but v_mfma_f32_4x4x4_16b_f16 computes 16 independent 4x4 blocks D = X * W + D per
//    instruction; with X = x' * I (lane i of a block supplies x' * e_i) and W = the four
//    gathered centroid halves of the block's four lanes (lane j supplies ITS four weights as
//    column j), D[i][j] = x' * W_j[i]: every lane accumulates x' times its own four weights
//    in fp32, i.e. four v_fma_mix_f32 become one MFMA that issues beside the VALU stream
//    (layout checked on hardware: tools/mfma_probe.hip).  The four lanes of a block must
//    share x', so a block is ONE input column of FOUR consecutive vector-rows:
//    lane = (block = column chunk, j = row).
//  * folded form (the default arithmetic): y = sum_g (c + r) * f16(s_g x_g) + sum_g b_g x_g; the
//    main and residual halves are separate MFMAs into the same accumulator, so c + r is never
//    formed: 4 v_perm_b32 + 4 MFMA per index instead of 14 VALU.  f16(s x) and sum b x are
//    computed ONCE per workgroup and staged in LDS.
//  * exact form: the reference's three roundings r16(r16(r16(c+r)*s)+b) stay packed-f16 VALU
//    (12 ops), the 8 fp32 FMAs become 2 MFMAs (f16 x f16 products are exact in fp32, so the
//    sum is the same fma chain): 16 VALU + 2 MFMA instead of 22 VALU.
//  * 1024-thread workgroups, one per CU (4 waves / SIMD).  The LDS image holds 8 replicas of
//    each codebook in one 256-byte row per entry (64 KiB); the two gathers of an index are
//    split across the lanes (bit 3 of the lane id picks main-then-residual or residual-then-
//    main), so every lane of a ds_read_b128 group reads its own bank quad -> conflict free
//    (4.8-5.1 LDS cycles per gather) with half the image of a 16-replica layout.
//  * persistent: one workgroup per CU (the CUs are shared out between the layers of a grouped
//    launch in proportion to their rows); it builds the image and stages the activations
//    once, then walks row groups bid, bid + wgs, ... of 4 vector-rows (32
//    outputs) each.  The packed index words stream through a register queue of NS slots (one
//    row group), D = 2 sweeps in flight: while sweep s is consumed, the sweep D positions
//    further down the stream - of this row group or the NEXT one - is requested, so HBM
//    never idles between row groups.  Every region between two waits is straight-line code
//    with a fixed, unconditional set of loads (no predicated load, no LDS-DMA) fenced by
//    sched_barriers: that is what lets the compiler emit the exact in-order count,
//    s_waitcnt vmcnt(D - 1), in the steady-state loop; any conditional load or visible LDS-DMA
//    in flight made it wait for ALL loads, i.e. serialised HBM latency with compute.
//  * 2-4 tokens per launch (TOK): the same MFMA as a real contraction over two columns and the
//    two tables, token = block row (sweep_tokens below).
//  * two entry points share the body: gemv_k256m_kernel (layer = blockIdx.y, grouped launches)
//    and gemv_k256m_kernel_1 (one layer, one token: arguments preloaded into SGPRs).
#include <type_traits>

#include "common.h"
#include "kernels.h"
#include "k256.h"

namespace vptq {

constexpr int kMThreads = 1024;
constexpr int kMWaves = kMThreads / 64;
constexpr int kMSweepCols = kMWaves * 16 * 8;  // 2048: 16 blocks (column chunks of 8) per wave
constexpr int kMTableBytes = 65536;  // 256 rows x (8 main + 8 residual replicas) x 16 B
constexpr int kMMaxLds = 163840;   // 160 KiB per CU
constexpr int kMMaxCols = 14336;   // staged activations must fit beside the image
constexpr int kMRedSlot = kMWaves * 32 * 4;   // one row group's cross-wave partials, per token
constexpr int kMMaxSlots = 4;
// VPTQ_GEMV_SELECTIVE (round 6; one token, fp16, folded instantiations; K256Layer::slots bit 8): the folded form, with the reference's
// roundings on the blocks of 128 columns an activation dominates - see gemv_k256c.hip.  Here everything happens inside the launch:
// while a wave stages its 512 columns it takes the rms of ITS f16(s x), marks the blocks that hold a column at or above
// kMSelKappa x that rms (one flag byte per block), and stages zeros for them (so the folded loop adds nothing there, and sum b x
// leaves them out); behind the prologue barrier the workgroup rebuilds the hot blocks' weights of its own row groups as the
// reference rounds them and leaves sum w x in `corr` (LDS), which finish() adds before the one rounding.  Nothing in the loop changes.
constexpr float kMSelKappa = 6.0f;
constexpr int kMSelBit = 0x100;            // K256Layer::slots: selective roundings
constexpr int kMSelMaxBlocks = 128;        // 16384 columns (staged: <= 14336)
constexpr int kMSelRowGroups = 16;         // row groups per workgroup the corrections cover (16 waves = 4 x 4 rows per round; gate / up of an
                                           // 8B model in one grouped launch: 14 per workgroup)
constexpr int kMSelMaxHot = 16;            // hot blocks per layer the corrections cover: the most dominant ones (a token with dozens of
                                           // "dominant" blocks is a dense one - the folded form's regime - and the corrections are latency-bound work: the
                                           // 14336-column down projection of a decoder, silu(gate) * up as input, had 50 of 112 blocks hot: 49 us for a 6 us launch)
constexpr int kMSelBytes = 2 * kMSelMaxBlocks + 4 * kMSelRowGroups * 32 * 4;   // block magnitudes + corr [4 quarters of the hot blocks][row groups][32]
// VPTQ_K256M_XDUP: one token - the staged activations are kept as DUPLICATED pairs (x, x), one
// dword per column, so that the MFMA x operand (x * e_j) is two v_and_b32 with per-lane
// constant masks instead of two v_perm_b32 (a three-source VOP3: ~5.5 vs ~3.3 cycles per
// wave-instruction, tools/ubench.hip).  Costs 2 more bytes of LDS per column.
#ifndef VPTQ_K256M_XDUP
#define VPTQ_K256M_XDUP 0
#endif
// VPTQ_K256M_SB_ALL (A/B): every exact instantiation stages scale and bias in LDS (default: bf16, and fp16 from 6 sweeps on)
#ifndef VPTQ_K256M_SB_ALL
#define VPTQ_K256M_SB_ALL 0
#endif
constexpr int kMXBytes1 = VPTQ_K256M_XDUP ? 4 : 2;  // staged bytes per column, one token

// Timing-only ablations (tools/gpu_ablate.sh; the RESULTS of such a build are wrong): bit 0 no
// MFMAs, bit 1 no LDS gathers (the addresses are still computed), bit 2 no epilogue, bit 3 no
// index loads, bit 4 only the two main-table MFMAs, bit 5 only one MFMA per index, bit 6 no bias
// load (sum b x), bit 7 no scale load.  What they
// measured (profiles/r02/k256m_ablation_*.txt): every instruction of the inner loop costs its
// issue time - the four 2-pass MFMAs 27-30 SIMD cycles per index-wave, the two gathers 12 - and
// little of it hides behind anything else.
#ifndef VPTQ_K256M_ABLATE
#define VPTQ_K256M_ABLATE 0
#endif
constexpr bool kAblNoMfma = (VPTQ_K256M_ABLATE & 1) != 0;
constexpr bool kAblNoGather = (VPTQ_K256M_ABLATE & 2) != 0;
constexpr bool kAblNoFinish = (VPTQ_K256M_ABLATE & 4) != 0;
constexpr bool kAblNoIndex = (VPTQ_K256M_ABLATE & 8) != 0;
constexpr bool kAblHalfMfma = (VPTQ_K256M_ABLATE & 16) != 0;
constexpr bool kAblQuarterMfma = (VPTQ_K256M_ABLATE & 32) != 0;
constexpr bool kAblNoBias = (VPTQ_K256M_ABLATE & 64) != 0;    // the per-column bias values are not loaded
constexpr bool kAblNoScale = (VPTQ_K256M_ABLATE & 128) != 0;  // the per-column scales are not loaded

static __device__ __forceinline__ u32x4 ldg16(const void* base, uint32_t byte_off) {
  return *(const u32x4*)as_global((const char*)base + byte_off);
}

// Queue load: 16 bytes at (wave-uniform base) + (32-bit lane offset).  An ordinary load whose
// wait the compiler places (see the file comment).  A version with inline-assembly loads and
// hand-written s_waitcnt counts was faster to write and unsafe: nothing stops the register
// allocator from copying a loaded register between the load and its wait, and the hardware
// does not interlock - it produced wrong sums on some runs.
static __device__ __forceinline__ void q_load(u32x4& dst, const void* sbase, uint32_t voff) {
  dst = *(const u32x4*)as_global((const char*)sbase + voff);
}
// The packed index words are read exactly once per launch, by one CU: non-temporal, so that
// they do not push the codebooks / activations / scales every workgroup re-reads out of L2
// (guide, "nt-weights": -5...10 % per decode layer).
#ifndef VPTQ_K256M_NT
#define VPTQ_K256M_NT 1
#endif
static __device__ __forceinline__ void q_load_stream(u32x4& dst, const void* sbase, uint32_t voff) {
#if VPTQ_K256M_NT
  dst = __builtin_nontemporal_load((const u32x4*)as_global((const char*)sbase + voff));
#else
  q_load(dst, sbase, voff);
#endif
}

// The kernel proper; Ly = this workgroup's layer, however its arguments arrived (see the two
// __global__ entry points below).
template <typename DT, int NS, int NST, bool PERM, bool FAST, int TOK>
static __device__ __forceinline__ void gemv_k256m_body(const K256Layer& Ly, const int tokens_arg) {
  const int tokens = tokens_arg & (kOutF32Bit - 1);
  const bool out_f32 = (tokens_arg & kOutF32Bit) != 0;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  {
    typedef __attribute__((address_space(3))) unsigned char lds_u8_t;
    if ((uint32_t)(uintptr_t)(lds_u8_t*)smem != 0u) __builtin_trap();  // absolute LDS addressing
  }
  // sweeps in flight per wave.  Bandwidth x latency is ~35 KB per CU; 2 sweeps (32 KiB) cover
  // it; a deeper queue in the steady state changes nothing (3 or 4 sweeps: same launch time,
  // same grouped time - the loop is bound by instruction issue, not by bytes in flight).
  // In the PROLOGUE every index load that is in flight before a wave's activations have arrived
  // delays them, and with them the barrier all 16 waves wait at: requesting the first sweep
  // right behind the image (DP = 1, the round-1 order had DP = 2) costs 0.2-0.5 us per launch
  // against requesting it once the wave has staged its activations (DS); two sweeps at that
  // point, or none before the barrier, are slower again (same-box A/B of all five orders:
  // profiles/r02/k256m_prologue_order_ab.txt; 8192^2 single launch 7.6-8.8 -> 7.0-7.7 us).
#ifndef VPTQ_K256M_DEPTH
#define VPTQ_K256M_DEPTH 2
#endif
#ifndef VPTQ_K256M_DEPTH_PROLOGUE
#define VPTQ_K256M_DEPTH_PROLOGUE 0
#endif
#ifndef VPTQ_K256M_DEPTH_STAGED
#define VPTQ_K256M_DEPTH_STAGED 1
#endif
  constexpr int D = NS < VPTQ_K256M_DEPTH ? NS : VPTQ_K256M_DEPTH;  // steady state
  constexpr int DP = D < VPTQ_K256M_DEPTH_PROLOGUE ? D : VPTQ_K256M_DEPTH_PROLOGUE;  // behind the image
  constexpr int DS = DP + VPTQ_K256M_DEPTH_STAGED < D ? DP + VPTQ_K256M_DEPTH_STAGED : D;  // + behind the staging
  // NST = 0: activations are NOT staged in LDS (more than 14336 columns do not fit beside the
  // image): the queue carries scale and x next to the index words, a row group is walked in
  // column blocks of NS sweeps, and the folded form only.
  constexpr bool STAGE = NST > 0;
  static_assert(STAGE || FAST, "the unstaged variant exists for the folded arithmetic only");
  // TOK = 2 / 4: up to TOK activation rows per launch (folded form, staged).  The MFMA is then a
  // real contraction - see sweep() - and costs the same for 2, 3 or 4 tokens.
  static_assert(TOK == 1 || TOK == 2 || TOK == 4, "token slots");
  static_assert(TOK == 1 || STAGE, "several tokens: staged activations");
  // bf16 exact form: scale and bias are staged in LDS beside the activations (6 bytes per column) instead of riding in the queue -
  // its matrix-pipe roundings (sweep()) need the registers
  // (fp16, 6 and 7 sweeps: the queued form spills; several tokens in the reference's roundings - round 6: the one-token loop with one
  // pair of MFMAs more per token - always staged)
  constexpr bool kSB = !FAST && (std::is_same<DT, BF16>::value || NS > 5 || TOK > 1 || VPTQ_K256M_SB_ALL);
  constexpr int NQ = ((FAST && STAGE) || kSB) ? 1 : NS;  // queue slots for scale + (exact: bias | unstaged: x)

  const int bid = blockIdx.x;
  const int G = Ly.G, N = Ly.N, O = Ly.O;
  const int n_groups = (N + kMRows - 1) / kMRows;
  const int step = Ly.wgs;  // this layer's share of the CUs (<= n_groups)
  if (bid >= step) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: an SGPR
  const int j = lane & 3;     // vector-row inside the group = position inside the MFMA block
  const int blk = lane >> 2;  // MFMA block = chunk of 8 columns
  const uint16_t* const sp = Ly.scale;
  const uint16_t* const bp = Ly.wbias;
  const uint32_t row_bytes = (uint32_t)Ly.row_words * 4u;

  K256_STAMP(kMWaves, 0, tid);

  // gather address = perm(word, base, sel): {0, base.b2 = table, index byte, base.b0 = replica}
  // Image row e = 8 replicas of main entry e (16-byte slots 0-7) + 8 replicas of residual entry
  // e (slots 8-15).  The two gathers of an index are split ACROSS the lanes: in gather A the
  // lanes with bit 3 clear fetch the main entry (slot lane & 7) and the others the residual
  // entry (slot 8 + (lane & 7)); gather B is the complement.  Every 16-lane group of a
  // ds_read_b128 then touches 16 different slots - conflict free with 64 KiB instead of the
  // 128 KiB a 16-replica image of both tables needs (tools/ubench_lds.hip: 4.8-5.1 cycles
  // either way) - and it does not matter which register holds which: both are accumulated
  // with the same x operand (exact form: added).
  const uint32_t hi = (lane >> 3) & 1u;
  const uint32_t baseA = ((hi << 3) | (lane & 7u)) << 4;
  const uint32_t baseB = (((hi ^ 1u) << 3) | (lane & 7u)) << 4;
  // selector {0, 0, word.byte[p], base.byte0}: index byte 2h is the main index of the word's
  // element h, byte 2h + 1 its residual index
  const uint32_t selGA[2] = {0x0c0c0400u | (hi << 8), 0x0c0c0600u | (hi << 8)};
  const uint32_t selGB[2] = {0x0c0c0400u | ((hi ^ 1u) << 8), 0x0c0c0600u | ((hi ^ 1u) << 8)};
  // x operand of the MFMA: x' * e_j as two packed pairs, cut out of a packed x' register
  // (column parity h picks the half) by one v_perm_b32 with a per-lane selector
  const uint32_t selA[2] = {j == 0 ? 0x0c0c0504u : j == 1 ? 0x05040c0cu : 0x0c0c0c0cu,
                            j == 0 ? 0x0c0c0706u : j == 1 ? 0x07060c0cu : 0x0c0c0c0cu};
  const uint32_t selB[2] = {j == 2 ? 0x0c0c0504u : j == 3 ? 0x05040c0cu : 0x0c0c0c0cu,
                            j == 2 ? 0x0c0c0706u : j == 3 ? 0x07060c0cu : 0x0c0c0c0cu};

  // LDS map: [0, 64 KiB) codebook image | per token: G staged activations + 8 zeros (the
  // operand of columns past G) + a 16-byte dump slot | cross-wave scratch
  const uint32_t xs_off = kMTableBytes;
  constexpr bool kXDup = VPTQ_K256M_XDUP && TOK == 1 && STAGE;
  constexpr uint32_t kXB = kXDup ? 4u : 2u;  // staged bytes per column
  const uint32_t xs_stride = (uint32_t)G * kXB + 32u;
  // bf16 exact form: the one-hot first operand e_j (1.0 at position j)
  const u32x2 bf_id = u32x2{__builtin_amdgcn_perm(0x3f803f80u, 0u, selA[0]), __builtin_amdgcn_perm(0x3f803f80u, 0u, selB[0])};
  const uint32_t maskA = j == 0 ? 0x0000ffffu : j == 1 ? 0xffff0000u : 0u;
  const uint32_t maskB = j == 2 ? 0x0000ffffu : j == 3 ? 0xffff0000u : 0u;
  const uint32_t sb_off = xs_off + (uint32_t)TOK * xs_stride;   // (kSB) scale plane, then the bias plane: the activations' layout
  const uint32_t red_off = xs_off + (STAGE ? TOK * xs_stride : 0u) + (kSB ? 2u * xs_stride : 0u);
  float* const red_b = (float*)(smem + red_off);        // [TOK][kMWaves]: sum b * x per wave
  uint32_t* const slot_cnt = (uint32_t*)(red_b + TOK * kMWaves);  // [kMMaxSlots] waves that have arrived
  uint32_t* const slot_done = slot_cnt + kMMaxSlots;         // [kMMaxSlots] row groups finished + 1
  // one token: hot-block flags and corrections of the selective arithmetic in front of the slots
  uint16_t* const hot_mag = (uint16_t*)(slot_done + kMMaxSlots + 8);    // [kMSelMaxBlocks]: largest |f16(s x)| of a block at or above its wave's threshold, else 0
  float* const corr = (float*)(hot_mag + kMSelMaxBlocks);               // [4][kMSelRowGroups][32]
  float* const red = (float*)(slot_done + kMMaxSlots + 8) + (TOK == 1 && !kSB ? kMSelBytes / 4 : 0);   // [K][TOK][kMWaves][32]
  const int K = Ly.slots & 0xff;  // partial-sum slots that fit into LDS (1..kMMaxSlots, host)
  constexpr bool kSelOk = FAST && (NST > 0) && TOK == 1;   // (fp16 and bf16: the corrections are VALU arithmetic in either type)
  const bool sel = kSelOk && (Ly.slots & kMSelBit) != 0;   // (wave-uniform)
  bool sel_any = false;                                     // ... and a block is hot: finish() adds the corrections
  uint32_t sel_mg[(NST > 0 ? NST : 1)];                     // largest magnitude of this thread's 8 columns per staged chunk

  // ---- 2. the index queue: slot s holds sweep s (2048 columns x 4 rows, 16 bytes per lane)
  u32x4 iw[NS], s_raw[NQ], b_raw[NQ];
  const int lane_cols = (wave * 16 + blk) * 8;  // this lane's 8 columns inside a sweep
  const int n_cblocks = STAGE ? 1 : (G + NS * kMSweepCols - 1) / (NS * kMSweepCols);
  auto issue_sweep = [&](int s, int rg, int cb) {
    // rows past N re-read the last row and are not stored; columns past G re-read the last 8
    const int row0 = rg * kMRows;
    const char* const rbase = (const char*)Ly.idx + (size_t)row0 * row_bytes;  // wave-uniform
    const uint32_t roff = (uint32_t)(row0 + j < N ? j : N - 1 - row0) * row_bytes;
    const int want = (cb * NS + s) * kMSweepCols + lane_cols;
    const uint32_t coff = (uint32_t)(want < G ? want : G - 8) * 2u;  // G % 8 == 0 (host check)
    if (!FAST && !kSB) {
      q_load(s_raw[NQ > 1 ? s : 0], sp, coff);
      q_load(b_raw[NQ > 1 ? s : 0], bp, coff);
    } else if (FAST && !STAGE) {
      q_load(s_raw[NQ > 1 ? s : 0], sp, coff);
      q_load(b_raw[NQ > 1 ? s : 0], Ly.x, coff);
    }
    if constexpr (kAblNoIndex) {
      iw[s] = u32x4{roff + coff, coff * 2654435761u, roff ^ 0x5a5a5a5au, coff + 0x01234567u};
      asm volatile("" : "+v"(iw[s]));
    } else {
      q_load_stream(iw[s], rbase, roff + coff);
    }
  };

  // ---- 3. prologue, once per workgroup.
  //  * LDS codebook image: thread t replicates entry (t >> 1) & 255 of table t >> 9 into 4 of
  //    its 8 slots (rotated so that the 8 lanes of a ds_write_b128 group hit 8 different
  //    slots).
  //  * activations: thread t stages columns 8t.. of each 8192-column block.  FAST stages
  //    f16(s * x) and sums b * x; the exact form stages x itself.
  // Order (tools/trace_k256m.py): the codebook entry and the activations are requested
  // together (two cold misses overlap), the image is written as soon as the entry is there,
  // the activations are staged, and only then does the index queue go out (DP / DS above) - a
  // wave issues in order, and behind a backed-up vector-memory queue the image and the
  // activations, which all 16 waves wait for, arrived 0.5-3 us later.  (An LDS-DMA fill,
  // global_load_lds_dwordx4, was slower still.)
  //  * sum b * x (one token, folded form): the bias values are requested behind the image write
  //    (round 1: behind the first index sweeps; with those now requested after the staging it is
  //    the same place) and the dot product waits until the first sweep has been consumed
  //    (late_bias below).  Loaded with the rest, these 16 KiB per workgroup - the same lines for all 256
  //    workgroups at the same moment, like x and scale - delayed the prologue barrier by
  //    0.6-0.7 us per launch (timing experiment without the load: 8.4 -> 7.8 us).
  constexpr int kStageCols = kMThreads * 8;
  constexpr int kSt = NST > 0 ? NST : 1;
  constexpr bool kLateB = FAST && STAGE && TOK == 1;
  u32x4 late_x[kLateB ? kSt : 1], late_b[kLateB ? kSt : 1];
  {
    u32x4 st_x[kSt][TOK], st_s[kSt], st_b[kSt], centry;
    u32x4 st_pv[PERM ? kSt : 1];  // PERM: 8 input-feature numbers (uint16) per staged chunk
    const char* const c0 = (const char*)Ly.cent;
    const uint32_t cent_off = (uint32_t)((tid >> 1) & 255) * 16u;
    const uint32_t rowp = ((uint32_t)((tid >> 1) & 255) << 8) | ((uint32_t)(tid >> 9) << 7) |
                          ((uint32_t)(tid & 1) << 6);
    auto write_image = [&]() {
#pragma unroll
      for (int q = 0; q < 4; ++q) lds_store16(rowp + (((q + (lane >> 1)) & 3) << 4), centry);
    };
    {
      // wave-uniform table choice: waves 0-7 replicate the main codebook, 8-15 the residual
      const char* const tab = __builtin_amdgcn_readfirstlane(wave) < 8 ? c0 : (const char*)Ly.rcent;
      q_load(centry, tab, cent_off);
#pragma unroll
      for (int k = 0; k < NST; ++k) {
        const int want = k * kStageCols + tid * 8;
        const int col0 = want < G ? want : G - 8;
        const uint32_t off = (uint32_t)col0 * 2u;
        if (PERM) q_load(st_pv[k], Ly.perm, off);
        if (FAST || kSB) {
          if constexpr (kAblNoScale) st_s[k] = u32x4{0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
          else q_load(st_s[k], sp, off);
          if (!kLateB) q_load(st_b[k], bp, off);
        }
        // PERM: x in its own order, permuted through LDS below.  Token slots past `tokens`
        // repeat the last row (their outputs are not stored).
#pragma unroll
        for (int t = 0; t < TOK; ++t)
          q_load(st_x[k][t], (const char*)Ly.x + (TOK == 1 ? (size_t)0 : (size_t)(t < tokens ? t : tokens - 1) * (size_t)G * 2u), off);
      }
      __builtin_amdgcn_sched_barrier(0);
      write_image();
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < DP; ++s) issue_sweep(s, bid, 0);
      if (kLateB) {
#pragma unroll
        for (int k = 0; k < NST; ++k) {
          const int want = k * kStageCols + tid * 8;
          if constexpr (kAblNoBias) late_b[k] = u32x4{0, 0, 0, 0};
          else q_load(late_b[k], bp, (uint32_t)(want < G ? want : G - 8) * 2u);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      K256_STAMP(kMWaves, 1, tid);
    }
    if (PERM) {
      // Column c of the quantised matrix multiplies input feature perm[c].  Gathering x through
      // the permutation from global memory is 8192 scattered 2-byte loads per workgroup
      // (+2.3 us); instead x is parked in the staging area in its own order and gathered from
      // LDS: store, barrier, 8 ds_read_u16 per thread, barrier, then staged like the rest.
#pragma unroll
      for (int k = 0; k < NST; ++k) {
        const int want = k * kStageCols + tid * 8;
#pragma unroll
        for (int t = 0; t < TOK; ++t)
          lds_store16(xs_off + t * xs_stride + (uint32_t)(want < G ? want : G + 8) * 2u, st_x[k][t]);
      }
      __syncthreads();
      typedef __attribute__((address_space(3))) const uint16_t lds_u16_t;
#pragma unroll
      for (int k = 0; k < NST; ++k)
#pragma unroll
        for (int t = 0; t < TOK; ++t)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const uint32_t xt = xs_off + t * xs_stride;
            const uint32_t lo = *(lds_u16_t*)(uintptr_t)(xt + (st_pv[k][q] & 0xffffu) * 2u);
            const uint32_t hi = *(lds_u16_t*)(uintptr_t)(xt + (st_pv[k][q] >> 16) * 2u);
            st_x[k][t][q] = lo | (hi << 16);
          }
      __syncthreads();  // every thread has its activations: the area may be overwritten
    }
    float accb[TOK];
#pragma unroll
    for (int t = 0; t < TOK; ++t) accb[t] = 0.f;
    u32x4 sel_xp[kSelOk ? kSt : 1];   // selective: this thread's staged f16(s x)
    K256_STAMP(kMWaves, 6, st_x[0][0][0]);  // (trace build) the activations have arrived
#pragma unroll
    for (int k = 0; k < NST; ++k) {
      const int want = k * kStageCols + tid * 8;
      const bool valid = want < G;
      // columns past G: x = 0 (adds nothing to sum b * x), the store goes to the dump slot
      const uint32_t keep = valid ? 0xffffffffu : 0u;
#pragma unroll
      for (int t = 0; t < TOK; ++t) {
        u32x4 v = st_x[k][t];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] &= keep;
        if (kLateB) late_x[k] = v;
        if (FAST) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            if (!kLateB) accb[t] = DT::dot2(v[q], st_b[k][q], accb[t]);  // sum b * x
            v[q] = DT::mul2(v[q], st_s[k][q]);                           // f16(s * x)
          }
          if constexpr (kSelOk) sel_xp[k] = v;
        }
        if constexpr (kXDup) {
          // (x_c, x_c) per column: 32 bytes per thread
          const uint32_t a = xs_off + (uint32_t)(valid ? want : G + 4) * 4u;
          lds_store16(a, u32x4{__builtin_amdgcn_perm(v[0], v[0], 0x01000100u), __builtin_amdgcn_perm(v[0], v[0], 0x03020302u),
                               __builtin_amdgcn_perm(v[1], v[1], 0x01000100u), __builtin_amdgcn_perm(v[1], v[1], 0x03020302u)});
          if (valid)
            lds_store16(a + 16u, u32x4{__builtin_amdgcn_perm(v[2], v[2], 0x01000100u), __builtin_amdgcn_perm(v[2], v[2], 0x03020302u),
                                       __builtin_amdgcn_perm(v[3], v[3], 0x01000100u), __builtin_amdgcn_perm(v[3], v[3], 0x03020302u)});
        } else {
          lds_store16(xs_off + t * xs_stride + (uint32_t)(valid ? want : G + 8) * 2u, v);
        }
      }
      if constexpr (kSB) {
        lds_store16(sb_off + (uint32_t)(valid ? want : G + 8) * 2u, st_s[k]);
        lds_store16(sb_off + xs_stride + (uint32_t)(valid ? want : G + 8) * 2u, st_b[k]);
      }
    }
    if constexpr (kSelOk) {
      if (sel) {
        // hot blocks: threshold = kMSelKappa x rms of f16(s x) over the 512 NST columns THIS WAVE stages (no barrier needed).
        // Columns past G count as zeros: a wave with a handful of valid columns gets a LOW threshold (its columns turn hot
        // sooner), never a useless one (a lone column is only sqrt(n) x the rms of n columns).
        float ss = 0.f;
        constexpr int cnt = 512 * (NST > 0 ? NST : 1);
#pragma unroll
        for (int k = 0; k < NST; ++k) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float a = DT::lo(sel_xp[k][q]), b = DT::hi(sel_xp[k][q]);
            ss = __builtin_fmaf(a, a, __builtin_fmaf(b, b, ss));
          }
        }
        ss = wave_sum(ss);
        uint32_t thr = DT::kInfBits;                  // inf / NaN sums: only inf / NaN columns are hot
        {
          const float tq = kMSelKappa * __builtin_sqrtf(ss / (float)cnt);
          if (tq < DT::kMaxFinite) {
            thr = (uint32_t)DT::from_float(tq) & 0x7fffu;
            if (DT::to_float((uint16_t)thr) < tq) thr += 1u;
          }
          if (thr == 0u) thr = 1u;                    // (all-zero activations: nothing is hot)
        }
        // largest magnitude of this thread's 8 columns per staged chunk; of its block of 128 columns (16 consecutive lanes): DPP max
#pragma unroll
        for (int k = 0; k < NST; ++k) {
          uint32_t mg = 0u;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const uint32_t lo = sel_xp[k][q] & 0x7fffu, hi = (sel_xp[k][q] >> 16) & 0x7fffu;
            mg = mg > lo ? mg : lo;
            mg = mg > hi ? mg : hi;
          }
          const int want = k * kStageCols + tid * 8;
          mg = want < G ? mg : 0u;
          uint32_t bm = mg;
          {
            uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)bm, 0x128, 0xf, 0xf, false);   // row_ror:8
            bm = bm > o ? bm : o;
            o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)bm, 0x124, 0xf, 0xf, false);            // row_ror:4
            bm = bm > o ? bm : o;
            o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)bm, 0x122, 0xf, 0xf, false);            // row_ror:2
            bm = bm > o ? bm : o;
            o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)bm, 0x121, 0xf, 0xf, false);            // row_ror:1
            bm = bm > o ? bm : o;
          }
          sel_mg[k] = bm;
          if (want < G && (lane & 15) == 0) hot_mag[want >> 7] = (uint16_t)(bm >= thr ? bm : 0u);   // one writer per block: no init, no atomics
        }
      }
    }
    if (!STAGE) {
      // the queue carries x and scale but not the bias: sum b * x in one pass over the columns
      for (int c = tid * 8; c < G; c += kStageCols) {
        const u32x4 xv = ldg16(Ly.x, (uint32_t)c * 2u), bv = ldg16(bp, (uint32_t)c * 2u);
#pragma unroll
        for (int q = 0; q < 4; ++q) accb[0] = DT::dot2(xv[q], bv[q], accb[0]);
      }
    }
    if (STAGE && tid < TOK) lds_store16(xs_off + tid * xs_stride + (uint32_t)G * kXB, u32x4{0, 0, 0, 0});
    if (kSB && tid < 2) lds_store16(sb_off + tid * xs_stride + (uint32_t)G * 2u, u32x4{0, 0, 0, 0});   // columns past G: weight 0
    if (tid < 2 * kMMaxSlots) slot_cnt[tid] = 0u;
    if (FAST && !kLateB) {
#pragma unroll
      for (int t = 0; t < TOK; ++t) {
        const float sum = wave_sum(accb[t]);
        if (lane == 0) red_b[t * kMWaves + wave] = sum;
      }
    }
  }
  // kLateB: this wave's share of sum b * x, once the bias values are there.  No barrier: the
  // value is read by the wave that arrives LAST at the first row group's slot counter, and
  // every wave writes it before it bumps that counter.
  auto late_bias = [&]() {
    float accb = 0.f;
#pragma unroll
    for (int k = 0; k < NST; ++k)
#pragma unroll
      for (int q = 0; q < 4; ++q) accb = DT::dot2(late_x[k][q], late_b[k][q], accb);
    const float sum = wave_sum(accb);
    if (lane == 0) red_b[wave] = sum;
  };
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int s = DP; s < DS; ++s) issue_sweep(s, bid, 0);  // this wave's activations are staged
  __builtin_amdgcn_sched_barrier(0);  // nothing that waits for index words above the barrier
  K256_STAMP(kMWaves, 7, tid);        // (trace build) this wave is at the barrier
  __syncthreads();
  __builtin_amdgcn_sched_barrier(0);
  K256_STAMP(kMWaves, 2, tid);
#ifdef VPTQ_K256_TRACE
  // (trace build) shader-clock cycles of the accumulate phase: with the wall-clock stamps 2 and 3
  // this gives the clock the CU actually ran at
  unsigned long long clk2_;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(clk2_) : "v"(tid) : "memory");
#endif
  // past the barrier the queue is filled to its steady-state depth
#pragma unroll
  for (int s = DS; s < D; ++s) issue_sweep(s, bid, 0);
  if constexpr (kSelOk) {
    if (sel) {
      const int nblk = (G + 127) >> 7;
      // every wave reads all block magnitudes (lane i: blocks i and i + 64) and keeps the kMSelMaxHot largest: the cut is found by
      // bisection on the 15 magnitude bits (ballots + popcounts: scalar work), the same in every wave
      const uint32_t g0 = lane < nblk ? hot_mag[lane] : 0u, g1 = lane + 64 < nblk ? hot_mag[lane + 64] : 0u;
      uint32_t cut = 1u;
      {
        auto above = [&](uint32_t t) {
          return __builtin_popcountll(__builtin_amdgcn_ballot_w64(g0 >= t)) + __builtin_popcountll(__builtin_amdgcn_ballot_w64(g1 >= t));
        };
        if (above(1u) > kMSelMaxHot) {
          uint32_t lo = 1u, hi = 0x8000u;    // above(lo) > K >= above(hi)
          while (hi - lo > 1u) {
            const uint32_t mid = (lo + hi) >> 1;
            if (above(mid) > kMSelMaxHot) lo = mid; else hi = mid;
          }
          cut = hi;
        }
      }
      const unsigned long long m0 = __builtin_amdgcn_ballot_w64(g0 >= cut);
      const unsigned long long m1 = __builtin_amdgcn_ballot_w64(g1 >= cut);
      // the staging threads take their hot blocks out of the folded loop (zeros in LDS, out of sum b x); then everybody waits once more
      {
#pragma unroll
        for (int k = 0; k < NST; ++k) {
          const int want = k * kStageCols + tid * 8;
          if (want < G) {
            const int blk = want >> 7;
            const bool hot = (((blk < 64 ? m0 : m1) >> (blk & 63)) & 1ull) != 0ull;
            if (hot) {
              lds_store16(xs_off + (uint32_t)want * kXB, u32x4{0u, 0u, 0u, 0u});
              if constexpr (kXDup) lds_store16(xs_off + (uint32_t)want * kXB + 16u, u32x4{0u, 0u, 0u, 0u});
              late_x[k] = u32x4{0u, 0u, 0u, 0u};
            }
          }
        }
      }
      __syncthreads();   // (selective launches only)
      sel_any = (m0 | m1) != 0ull;   // (every wave reads the same flags: uniform over the workgroup)
      if (sel_any) {
        // wave w = (row group qi = w >> 2 of this workgroup, vector-row j = w & 3): the hot blocks' columns of that row, two per
        // lane, with the reference's roundings - w = f16(f16(f16(c + r) s) + b), vptq/ops/quant_gemm.py:121,155-156 - entries out of
        // the image (any replica: slot lane & 7), fp32 multiply-adds, one DPP sum per output.  Rare, short, not tuned.
        // all 16 waves: wave w = (vector-row jr = w & 3, quarter w >> 2 of the hot blocks); the quarters' sums meet in finish(), in a
        // fixed order.  Two hot blocks per trip: their index words, activations, scales and biases are requested together.
        const int jr = wave & 3, quad = wave >> 2;
        auto one_block = [&](const char* irow, int c, float (&ac)[8], bool on) __attribute__((always_inline)) {
          const int cc = (on && c < G) ? c : 0;       // (G is a multiple of 8: whole pairs; an absent second block reads column 0)
          const uint32_t iwd = *(const uint32_t*)as_global(irow + (size_t)cc * 2);
          uint32_t xw;
          if (PERM) {
            const uint32_t pv = *(const uint32_t*)as_global(Ly.perm + cc);
            xw = (uint32_t)as_global(Ly.x)[pv & 0xffffu] | ((uint32_t)as_global(Ly.x)[pv >> 16] << 16);
          } else {
            xw = *(const uint32_t*)as_global(Ly.x + cc);
          }
          if (!(on && c < G)) xw = 0u;                 // (... times zero)
          const uint32_t sw = *(const uint32_t*)as_global(sp + cc), bw = *(const uint32_t*)as_global(bp + cc);
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const uint32_t e = (iwd >> (16 * h)) & 0xffffu;
            const u32x4 ce = lds_load16((e & 255u) * 256u + ((uint32_t)lane & 7u) * 16u);
            const u32x4 re = lds_load16((e >> 8) * 256u + (8u + ((uint32_t)lane & 7u)) * 16u);
            const float xf = DT::half_of(xw, h);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              uint32_t w = DT::add2(ce[k], re[k]);
              w = DT::mul2_bcast(w, sw, h);
              w = DT::add2_bcast(w, bw, h);
              ac[2 * k] = DT::fma_lo(w, xf, ac[2 * k]);
              ac[2 * k + 1] = DT::fma_hi(w, xf, ac[2 * k + 1]);
            }
          }
        };
        // this quarter's hot blocks: every fourth set bit of (m0, m1)
        int mine[kMSelMaxBlocks / 4];
        int n_mine = 0;
        {
          int h = 0;
#pragma unroll 1
          for (int half = 0; half < 2; ++half) {
            unsigned long long m = half ? m1 : m0;
            while (m) {
              const int bit = __builtin_ctzll(m);
              m &= m - 1ull;
              if ((h & 3) == quad && n_mine < kMSelMaxBlocks / 4) mine[n_mine++] = half * 64 + bit;
              ++h;
            }
          }
        }
#pragma unroll 1
        for (int qi = 0; qi < kMSelRowGroups && bid + qi * step < n_groups; ++qi) {
          const int rgq = bid + qi * step;
          const int rowq = rgq * kMRows + jr;
          const char* const irow = (const char*)Ly.idx + (size_t)(rowq < N ? rowq : N - 1) * row_bytes;
          float ac[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
          for (int i = 0; i < n_mine; i += 2) {
            one_block(irow, mine[i] * 128 + 2 * lane, ac, true);
            one_block(irow, (i + 1 < n_mine ? mine[i + 1] : 0) * 128 + 2 * lane, ac, i + 1 < n_mine);
          }
#pragma unroll
          for (int k = 0; k < 8; ++k) ac[k] = wave_sum(ac[k]);
          if (lane == 0) {
#pragma unroll
            for (int k = 0; k < 8; ++k) corr[(quad * kMSelRowGroups + qi) * 32 + jr * 8 + k] = ac[k];
          }
        }
        __syncthreads();   // (uniform: every wave took this branch)
      }
    }
  }

  // ---- 4. row groups ----
  // one sweep: 8 indices per lane (2 gathers each, kAhead indices ahead of the arithmetic)
  // The MFMA x operands depend on the column only, not on the row group: a workgroup that
  // walks several row groups builds them during the first one (2 v_perm_b32 per index) and
  // keeps them in registers (16 per sweep) - afterwards an index costs 2 perms + 4 MFMA.
  // Only while they fit beside the rest (<= 4 sweeps, folded form).
#ifndef VPTQ_K256M_CACHE_XO
#define VPTQ_K256M_CACHE_XO 1
#endif
  constexpr bool kCacheXo = VPTQ_K256M_CACHE_XO && FAST && STAGE && NS <= 4 && TOK == 1;
  u32x2 xo_cache[kCacheXo ? NS : 1][8];
  // one token: acc.a[0 / 1][i] = output 0-3 / 4-7 of this lane's vector-row; several tokens:
  // acc.a[e][t] = output e of this lane's vector-row for token t
  // VPTQ_K256M_ACC4: the four MFMAs of an index go to four accumulators (main / residual x
  // low / high half), joined in finish(): no MFMA reads the result of the one two places
  // before it.  VPTQ_K256M_ADDFORM: c + r by four v_pk_add_f16 (the reference's first rounding),
  // then two MFMAs.
#ifndef VPTQ_K256M_ACC4
#define VPTQ_K256M_ACC4 0
#endif
#ifndef VPTQ_K256M_ADDFORM
#define VPTQ_K256M_ADDFORM 0
#endif
  constexpr bool kAcc4 = VPTQ_K256M_ACC4 && FAST && TOK == 1 && !VPTQ_K256M_ADDFORM;
  // (several tokens, reference roundings: a[2 t] / a[2 t + 1] = outputs 0-3 / 4-7 of token t, the one-token layout per token)
  constexpr int kNAcc = TOK == 1 ? (kAcc4 ? 4 : 2) : (FAST ? 8 : 2 * TOK);
  struct Acc { f32x4 a[kNAcc]; };
  // Several tokens (TOK = 2 / 4): here the 4x4x4 MFMA is a real contraction.  Lane i of a block
  // supplies token i's f16(s * x) of TWO columns, twice (A row i = {x'[c0], x'[c1], x'[c0],
  // x'[c1]}); lane j supplies, for ONE output element e, {W_A[c0][e], W_A[c1][e], W_B[c0][e],
  // W_B[c1][e]} of ITS vector-row (W_A / W_B = the two gathers of an index, i.e. main and
  // residual entry in either order).  D[i][j] += x'_i[c0] (c + r)[c0][e] + x'_i[c1] (c + r)[c1][e]:
  // lane j accumulates output e of its row for all four tokens.  The gathered entries hold 8
  // elements of one column, the operand wants one element of 2 columns: 8 v_perm_b32 per
  // index transpose them; with the 2 address perms that is 10 VALU + 4 MFMA per index for up to
  // four tokens (one token: 2-4 VALU + 4 MFMA).
  auto sweep_tokens = [&](int s, Acc& acc) {
    const int want = s * kMSweepCols + lane_cols;
    // this lane supplies the activations of token j (slots past TOK: the last one; unused rows)
    const uint32_t xt = xs_off + (uint32_t)(j < TOK ? j : TOK - 1) * xs_stride;
    const u32x4 xq = lds_load16(xt + (uint32_t)(want < G ? want : G) * 2u);  // past G: zeros
    const u32x4 words = iw[s];
    u32x4 ga[2][2], gb[2][2];  // [pair parity][column of the pair]
    auto gather_pair = [&](int p) {
      // (the empty asm keeps the compiler from deriving all 16 gather addresses of the sweep ahead
      // of the first fence: with 8 accumulators they no longer fit, it spilled them, and every
      // scratch reload waited with vmcnt(0), i.e. for the index sweep requested just before)
      uint32_t w = words[p];
      asm volatile("" : "+v"(w));
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        ga[p & 1][h] = lds_load16(__builtin_amdgcn_perm(w, baseA, selGA[h]));
        gb[p & 1][h] = lds_load16(__builtin_amdgcn_perm(w, baseB, selGB[h]));
      }
    };
    gather_pair(0);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      __builtin_amdgcn_sched_barrier(0);
      if (p + 1 < 4) gather_pair(p + 1);
      __builtin_amdgcn_sched_barrier(0);
      const u32x2 xa = u32x2{xq[p], xq[p]};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint32_t a0 = ga[p & 1][0][q], a1 = ga[p & 1][1][q];
        const uint32_t b0 = gb[p & 1][0][q], b1 = gb[p & 1][1][q];
        // elements 2q (low halves) and 2q + 1 (high halves) of the pair's two columns
        const u32x2 wlo = u32x2{__builtin_amdgcn_perm(a1, a0, 0x05040100u),
                                __builtin_amdgcn_perm(b1, b0, 0x05040100u)};
        const u32x2 whi = u32x2{__builtin_amdgcn_perm(a1, a0, 0x07060302u),
                                __builtin_amdgcn_perm(b1, b0, 0x07060302u)};
        constexpr int kE = (TOK == 1 || !FAST) ? 0 : 2;  // (never run with one token / in the reference's roundings)
        acc.a[kE * q] = DT::mfma4(xa, wlo, acc.a[kE * q]);
        acc.a[kE * q + kE / 2] = DT::mfma4(xa, whi, acc.a[kE * q + kE / 2]);
        // (fenced: the scheduler would transpose the whole pair first - 12 more live registers)
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  auto sweep = [&](auto first_c, int s, int cb, Acc& acc) {
    if constexpr (TOK > 1 && FAST) {
      sweep_tokens(s, acc);
      return;
    }
    f32x4& acc0 = acc.a[0];
    f32x4& acc1 = acc.a[1];
    constexpr bool kBuild = !kCacheXo || decltype(first_c)::value;
    const int want = (cb * NS + s) * kMSweepCols + lane_cols;
    u32x4 xq = u32x4{0, 0, 0, 0}, xq2 = u32x4{0, 0, 0, 0};
    [[maybe_unused]] u32x4 xqt[TOK > 1 ? TOK - 1 : 1];
    // (reference roundings, several tokens) the weight operands of an index serve every token: one more pair of MFMAs + 2 v_perm_b32 each
    auto more_tokens = [&](int q, int h, u32x2 wa, u32x2 wb) {
      if constexpr (TOK > 1) {
#pragma unroll
        for (int t = 1; t < TOK; ++t) {
          const u32x2 xt = u32x2{__builtin_amdgcn_perm(xqt[t - 1][q], 0u, selA[h]), __builtin_amdgcn_perm(xqt[t - 1][q], 0u, selB[h])};
          acc.a[2 * t] = DT::mfma4(xt, wa, acc.a[2 * t]);
          acc.a[2 * t + 1] = DT::mfma4(xt, wb, acc.a[2 * t + 1]);
        }
      }
    };
    if (STAGE) {
      if (kBuild) {
        xq = lds_load16(xs_off + (uint32_t)(want < G ? want : G) * kXB);  // past G: zeros
        if (kXDup) xq2 = want < G ? lds_load16(xs_off + (uint32_t)want * 4u + 16u) : xq;
      }
      if constexpr (TOK > 1) {   // (reference roundings) the further tokens' activations of these 8 columns
#pragma unroll
        for (int t = 1; t < TOK; ++t) xqt[t - 1] = lds_load16(xs_off + (uint32_t)t * xs_stride + (uint32_t)(want < G ? want : G) * kXB);
      }
    } else {
      const uint32_t keep = want < G ? 0xffffffffu : 0u;  // columns past G contribute 0
#pragma unroll
      for (int q = 0; q < 4; ++q)
        xq[q] = DT::mul2(b_raw[NQ > 1 ? s : 0][q] & keep, s_raw[NQ > 1 ? s : 0][q]);  // f16(s * x)
    }
    const u32x4 words = iw[s];
#ifndef VPTQ_K256M_AHEAD
#define VPTQ_K256M_AHEAD 2
#endif
    // indices whose gathers are in flight ahead of the MFMAs (the exact form has 12 more
    // registers of scale / bias per sweep in its queue: one less)
#ifndef VPTQ_K256M_SB_UNFENCED
#define VPTQ_K256M_SB_UNFENCED 0
#endif
#ifndef VPTQ_K256M_AHEAD_SB
#define VPTQ_K256M_AHEAD_SB 1
#endif
    constexpr int kAhead = FAST ? VPTQ_K256M_AHEAD : kSB ? VPTQ_K256M_AHEAD_SB : VPTQ_K256M_AHEAD - 1;
    u32x4 cv[kAhead + 1], rv[kAhead + 1];
    auto gather = [&](int u) {
      const uint32_t w = words[u >> 1];
      const int h = u & 1;
      const uint32_t aC = __builtin_amdgcn_perm(w, baseA, selGA[h]);
      const uint32_t aR = __builtin_amdgcn_perm(w, baseB, selGB[h]);
      if constexpr (kAblNoGather) {
        asm volatile("" :: "v"(aC), "v"(aR));
        cv[u % (kAhead + 1)] = words;
        rv[u % (kAhead + 1)] = words;
      } else {
        cv[u % (kAhead + 1)] = lds_load16(aC);
        rv[u % (kAhead + 1)] = lds_load16(aR);
      }
    };
#pragma unroll
    for (int u = 0; u < kAhead; ++u) gather(u);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      // fenced: left alone, the scheduler sinks the gathers next to their use and the loop
      // runs with one gather in flight (s_waitcnt lgkmcnt(1)), i.e. at LDS latency
#ifndef VPTQ_K256M_FENCE_UNITS
#define VPTQ_K256M_FENCE_UNITS 1
#endif
      if (VPTQ_K256M_FENCE_UNITS && !(kSB && VPTQ_K256M_SB_UNFENCED)) __builtin_amdgcn_sched_barrier(0);
      if (u + kAhead < 8) gather(u + kAhead);
      if (VPTQ_K256M_FENCE_UNITS && !(kSB && VPTQ_K256M_SB_UNFENCED)) __builtin_amdgcn_sched_barrier(0);
      const int q = u >> 1, h = u & 1;
      const u32x4 c = cv[u % (kAhead + 1)], r = rv[u % (kAhead + 1)];
      u32x2 xo;
      if (kBuild) {
        if constexpr (kXDup) {
          const uint32_t xd = u < 4 ? xq[u & 3] : xq2[u & 3];  // (x_u, x_u)
          xo = u32x2{xd & maskA, xd & maskB};
        } else {
          xo = u32x2{__builtin_amdgcn_perm(xq[q], 0u, selA[h]),
                     __builtin_amdgcn_perm(xq[q], 0u, selB[h])};
        }
        if (kCacheXo) xo_cache[kCacheXo ? s : 0][u] = xo;
      } else {
        xo = xo_cache[kCacheXo ? s : 0][u];
      }
      if (FAST && VPTQ_K256M_ADDFORM && std::is_same<DT, F16>::value) {
        uint32_t w2[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) w2[p] = DT::add2(c[p], r[p]);
        acc0 = DT::mfma4(xo, u32x2{w2[0], w2[1]}, acc0);
        acc1 = DT::mfma4(xo, u32x2{w2[2], w2[3]}, acc1);
      } else if (FAST && kAblNoMfma) {
        asm volatile("" :: "v"(c), "v"(r), "v"(xo));
      } else if (FAST) {
        f32x4& acc2 = acc.a[kAcc4 ? 2 : 0];
        f32x4& acc3 = acc.a[kAcc4 ? 3 : 1];
        acc0 = DT::mfma4(xo, u32x2{c[0], c[1]}, acc0);
        if constexpr (!kAblQuarterMfma) acc1 = DT::mfma4(xo, u32x2{c[2], c[3]}, acc1);
        if constexpr (!kAblHalfMfma && !kAblQuarterMfma) {
          acc2 = DT::mfma4(xo, u32x2{r[0], r[1]}, acc2);
          acc3 = DT::mfma4(xo, u32x2{r[2], r[3]}, acc3);
        } else {
          asm volatile("" :: "v"(r), "v"(c));
        }
      } else {
        u32x4 sv = s_raw[NQ > 1 ? s : 0], bv = b_raw[NQ > 1 ? s : 0];
        if constexpr (kSB) {
          if (u == 0) {
            s_raw[0] = lds_load16(sb_off + (uint32_t)(want < G ? want : G) * 2u);
            b_raw[0] = lds_load16(sb_off + xs_stride + (uint32_t)(want < G ? want : G) * 2u);
          }
          sv = s_raw[0];
          bv = b_raw[0];
        }
        if constexpr (std::is_same<DT, BF16>::value) {
          // bf16 (round 6): the reference's roundings are torch's bf16 ops - widen to fp32, operate, round - which on the VALU is
          // unpack / fp32 op / v_cvt_pk_bf16_f32 per stage: ~68 instructions per index, 21 us per 8192^2 layer (gemv_k256.hip).  The
          // matrix pipe widens for free: with the one-hot first operand e_j, lane j receives ITS OWN four second-operand values as
          // fp32 (D[i][j] = sum_k I[i][k] B[k][j]); accumulating a second such product adds in fp32, a first operand s e_j multiplies.
          // Every intermediate is a sum or a product of two bf16 values - exactly representable in fp32 or trivially rounded - so the
          // results are the widened arithmetic's bit for bit (tools/mfma_bf16_exact_probe.hip: 0 of 4 M values differ on
          // checkpoint-like, reference-test, wide-exponent and tie operands; an INFINITE or NaN weight turns its lane's other three
          // values into NaN as well: 0 x inf).  11 MFMAs + 12 conversions + 8 v_perm_b32 per index; 8192^2: 21.7 -> 11.9 us per launch, 9.98 us
          // per layer in a grouped launch.  Counters (profiles/r06/bf16_exact_mfma_pmc.json): the matrix pipe is 42 % busy and the time is
          // 8 cycles per MFMA + 4 per VALU instruction, added up - an A/B without the per-unit fences and with a deeper gather lookahead
          // (VPTQ_K256M_SB_UNFENCED, VPTQ_K256M_AHEAD_SB) changes nothing: issue bound, not latency bound.
          const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
          const u32x2 so = u32x2{__builtin_amdgcn_perm(sv[q], 0u, selA[h]), __builtin_amdgcn_perm(sv[q], 0u, selB[h])};   // s e_j
          f32x4 d0 = DT::mfma4(bf_id, u32x2{c[0], c[1]}, z);
          f32x4 d1 = DT::mfma4(bf_id, u32x2{c[2], c[3]}, z);
          d0 = DT::mfma4(bf_id, u32x2{r[0], r[1]}, d0);                                       // c + r, fp32
          d1 = DT::mfma4(bf_id, u32x2{r[2], r[3]}, d1);
          // b in every register: first operand = the bias of four columns as loaded (the same in all four lanes of a block), second
          // operand = the one-hot that picks this column: no VALU work
          const u32x2 one_hot = u32x2{(u & 3) == 0 ? 0x00003f80u : (u & 3) == 1 ? 0x3f800000u : 0u,
                                      (u & 3) == 2 ? 0x00003f80u : (u & 3) == 3 ? 0x3f800000u : 0u};
          const f32x4 bvec = DT::mfma4(u32x2{bv[(u >> 2) * 2], bv[(u >> 2) * 2 + 1]}, one_hot, z);
          u32x2 wa = u32x2{DT::pack(d0[0], d0[1]), DT::pack(d0[2], d0[3])};                   // first rounding
          u32x2 wb = u32x2{DT::pack(d1[0], d1[1]), DT::pack(d1[2], d1[3])};
          d0 = DT::mfma4(so, wa, z);                                                          // * s: exact products
          d1 = DT::mfma4(so, wb, z);
          wa = u32x2{DT::pack(d0[0], d0[1]), DT::pack(d0[2], d0[3])};                         // second rounding
          wb = u32x2{DT::pack(d1[0], d1[1]), DT::pack(d1[2], d1[3])};
          d0 = DT::mfma4(bf_id, wa, bvec);                                                    // + b, fp32
          d1 = DT::mfma4(bf_id, wb, bvec);
          wa = u32x2{DT::pack(d0[0], d0[1]), DT::pack(d0[2], d0[3])};                         // third rounding
          wb = u32x2{DT::pack(d1[0], d1[1]), DT::pack(d1[2], d1[3])};
          acc0 = DT::mfma4(xo, wa, acc0);
          acc1 = DT::mfma4(xo, wb, acc1);
          more_tokens(q, h, wa, wb);
          continue;
        }
        uint32_t w2[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) w2[p] = DT::add2(c[p], r[p]);
#pragma unroll
        for (int p = 0; p < 4; ++p) w2[p] = DT::mul2_bcast(w2[p], sv[q], h);
#pragma unroll
        for (int p = 0; p < 4; ++p) w2[p] = DT::add2_bcast(w2[p], bv[q], h);
        acc0 = DT::mfma4(xo, u32x2{w2[0], w2[1]}, acc0);
        acc1 = DT::mfma4(xo, u32x2{w2[2], w2[3]}, acc1);
        more_tokens(q, h, u32x2{w2[0], w2[1]}, u32x2{w2[2], w2[3]});
      }
    }
  };
  // reduce over the 16 column blocks of the wave, then over the 16 waves, and store.
  // lane (blk, j) holds 8 partial outputs (t = 0..7) of vector-row j.  Lane bits 5 and 4 by
  // swap-and-add (halving the values carried), bits 3 and 2 by DPP row rotations, which keep
  // lane & 3: afterwards lane l holds outputs 4*bit5 + 2*bit4 + {0, 1} of row l & 3.
  // Across the waves WITHOUT a barrier: a barrier per row group makes every wave wait for the
  // slowest one, and the SIMDs serve their waves oldest-first, so the waves of a workgroup
  // finish a row group more than a microsecond apart.  Instead each wave drops its 32 partials
  // into slot q % K (q = how many row groups this workgroup has finished) and bumps the slot's
  // LDS counter; the wave that arrives last sums the 16 x 32 partials, stores the 32 outputs
  // and releases the slot.  A wave only waits when it is K row groups ahead of the slowest.
  auto finish = [&](int rg, int q, const Acc& acc) {
    // v[e * TOK + t]: output e of this lane's vector-row, token t
    constexpr int NV = 8 * TOK;
    float v[NV];
    if constexpr (TOK == 1) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        v[i] = kAcc4 ? acc.a[0][i] + acc.a[kAcc4 ? 2 : 0][i] : acc.a[0][i];
        v[4 + i] = kAcc4 ? acc.a[1][i] + acc.a[kAcc4 ? 3 : 1][i] : acc.a[1][i];
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int t = 0; t < TOK; ++t) v[e * TOK + t] = FAST ? acc.a[FAST ? e : 0][t] : acc.a[FAST ? 0 : 2 * t + (e >> 2)][e & 3];
    }
#pragma unroll
    for (int i = 0; i < NV / 2; ++i) {
      auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[i]), __float_as_uint(v[i + NV / 2]),
                                                false, false);
      v[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
#pragma unroll
    for (int i = 0; i < NV / 4; ++i) {
      auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[i]), __float_as_uint(v[i + NV / 4]),
                                                false, false);
      v[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
#pragma unroll
    for (int i = 0; i < NV / 4; ++i) v[i] = row_ror_add<4>(row_ror_add<8>(v[i]));
    const int slot = q % K;
    if (q >= K) {
      // the slot's previous user (row group number q - K of this workgroup) must be stored
      while (__hip_atomic_load(&slot_done[slot], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) <
             (uint32_t)(q - K + 1))
        __builtin_amdgcn_s_sleep(2);
    }
    float* const rs = red + slot * (TOK * kMWaves * 32);  // [TOK][kMWaves][32]
    if ((lane & 12) == 0) {
      const int o8 = ((lane >> 5) & 1) * 4 + ((lane >> 4) & 1) * 2;
#pragma unroll
      for (int t = 0; t < TOK; ++t) {
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        *(f32x2*)&rs[(t * kMWaves + wave) * 32 + j * 8 + o8] = f32x2{v[t], v[TOK + t]};  // ds_write_b64
      }
    }
    uint32_t arrived = 0;
    if (lane == 0)
      arrived = __hip_atomic_fetch_add(&slot_cnt[slot], 1u, __ATOMIC_ACQ_REL,
                                       __HIP_MEMORY_SCOPE_WORKGROUP);
    arrived = __builtin_amdgcn_readfirstlane(arrived);
    if (arrived == kMWaves - 1) {  // last wave of this row group: wave-uniform branch
      // This is the tail of the kernel when the row group is the workgroup's last one: keep the
      // dependent chain short.  All 64 lanes: lane (half, o) sums 8 of the 16 waves' partials of
      // output o as a tree (8 LDS reads in flight, 3 add levels), one lane swap joins the halves;
      // sum b * x over the waves is a 16-lane DPP sum.  (A serial 32-add chain in 32 lanes cost
      // ~0.3 us per launch.)
      static_assert(kMWaves == 16, "final sum: 2 halves x 8 waves");
      // (the lane number afresh: the copy made at kernel start would have to live in a register,
      // or in scratch, until here)
      const int ln = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
      const int half = ln >> 5, ol = ln & 31;
      const int row = rg * kMRows + (ol >> 3);
      const int o = row * 8 + (ol & 7);
      const bool store = ln < 32 && row < N && o < O;
      float sum[TOK];
#pragma unroll
      for (int t = 0; t < TOK; ++t) {
        const float* const ps = rs + (t * kMWaves + half * 8) * 32 + ol;
        const float s0 = (ps[0] + ps[32]) + (ps[64] + ps[96]);
        const float s1 = (ps[128] + ps[160]) + (ps[192] + ps[224]);
        sum[t] = s0 + s1;
      }
      float bdot[TOK];  // sum b * x: the 16 waves' shares, added up in every 16-lane row
#pragma unroll
      for (int t = 0; t < TOK; ++t) bdot[t] = FAST ? row16_allsum(red_b[t * kMWaves + (ln & 15)]) : 0.f;
      float bv = 0.f;
      if (store && Ly.bias) bv = DT::to_float(as_global(Ly.bias)[o]);
#pragma unroll
      for (int t = 0; t < TOK; ++t) {
        auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(sum[t]), __float_as_uint(sum[t]), false, false);
        float total = (__uint_as_float(r[0]) + __uint_as_float(r[1])) + bdot[t];
        if constexpr (kSelOk) {
          if (sel_any) {   // (the hot blocks' columns, reference roundings: four quarters, a fixed order)
            const float* const cq = corr + (q < kMSelRowGroups ? q : 0) * 32 + ol;
            total += (cq[0] + cq[kMSelRowGroups * 32]) + (cq[2 * kMSelRowGroups * 32] + cq[3 * kMSelRowGroups * 32]);
          }
        }
        if (store && t < tokens) {
          if (out_f32) ((float*)as_global(Ly.y))[(size_t)t * O + o] = total + bv;
          else as_global(Ly.y)[(size_t)t * O + o] = DT::from_float(total + bv);
        }
      }
      if (lane == 0) {
        slot_cnt[slot] = 0u;
        __hip_atomic_store(&slot_done[slot], (uint32_t)(q + 1), __ATOMIC_RELEASE,
                           __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
  };

  // One row group = n_cblocks column blocks of NS sweeps (one block when the activations are
  // staged).  LAST = no further row group for this workgroup (the queue drains in its last
  // block); otherwise, when sweep s is consumed, the sweep D positions further down the
  // stream - same block, the next block, or the next row group - is requested into its slot,
  // so D - 1 younger sweeps are in flight behind the one being waited for.  The sched_barriers
  // keep load issue, gathers and arithmetic in this order (without them the scheduler hoists
  // loads and gathers until the kernel spills).
  auto cblock = [&](auto first_c, auto last_c, int rg, int cb, int next_rg, int next_cb, Acc& acc) {
    constexpr bool LAST = decltype(last_c)::value;
#define K256M_STEP(S)                                                                          \
  if constexpr (S < NS) {                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                         \
    sweep(first_c, S, cb, acc);                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                         \
    if constexpr (S + D < NS) issue_sweep(S + D, rg, cb);                                      \
    else if constexpr (!LAST) issue_sweep(S + D - NS, next_rg, next_cb);                       \
    __builtin_amdgcn_sched_barrier(0);                                                         \
    if constexpr (S == 0 && kLateB && decltype(first_c)::value) {                              \
      late_bias();                                                                             \
      __builtin_amdgcn_sched_barrier(0);                                                       \
    }                                                                                          \
  }
    K256M_STEP(0) K256M_STEP(1) K256M_STEP(2) K256M_STEP(3)
    K256M_STEP(4) K256M_STEP(5) K256M_STEP(6)
#undef K256M_STEP
  };
  using no_t = std::integral_constant<bool, false>;
  auto row_group = [&](auto first_c, auto last_c, int rg, int q) {
    constexpr bool LAST = decltype(last_c)::value;
    Acc acc;
#pragma unroll
    for (int i = 0; i < kNAcc; ++i) acc.a[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int cb = 0; cb + 1 < n_cblocks; ++cb) cblock(first_c, no_t{}, rg, cb, rg, cb + 1, acc);
    cblock(first_c, last_c, rg, n_cblocks - 1, rg + step, 0, acc);
    if (LAST) K256_STAMP(kMWaves, 3, acc.a[0][0] + acc.a[1][0]);
#ifdef VPTQ_K256_TRACE
    if (LAST) {
      unsigned long long clk3_;
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(clk3_) : "v"(acc.a[0][0] + acc.a[1][0]) : "memory");
      if (lane == 0) ((unsigned long long*)Ly.pf)[((size_t)bid * kMWaves + wave) * 8 + 4] = clk3_ - clk2_;
    }
#endif
    if constexpr (kAblNoFinish) {
      if (acc.a[0][0] + acc.a[1][3] == 1234.5f) as_global(Ly.y)[tid] = 1;
    } else {
      finish(rg, q, acc);
    }
  };
  using yes = std::integral_constant<bool, true>;
  using no = std::integral_constant<bool, false>;
  int rg = bid;
  if (rg + step >= n_groups) {
    row_group(yes{}, yes{}, rg, 0);
  } else {
    row_group(yes{}, no{}, rg, 0);
    int q = 1;
    for (rg += step; rg + step < n_groups; rg += step, ++q) row_group(no{}, no{}, rg, q);
    row_group(no{}, yes{}, rg, q);
  }
  K256_STAMP(kMWaves, 5, tid);

#ifndef VPTQ_K256_TRACE
  {
    // read-ahead of the next launch's index stream (see gemv_k256.hip), once the own stream
    // is done; nothing waits for it (its value is used only in a branch that is never taken)
    const long long want = (long long)bid * Ly.pf_chunk + (long long)tid * 128;
    const bool in = tid * 128 < Ly.pf_len && want + 4 <= Ly.pf_bytes;
    const char* pa = in ? Ly.pf + want : (const char*)Ly.cent;
    const uint32_t pf_word = *(const uint32_t*)as_global(pa);
    if (tokens == 0x7fffffff) as_global(Ly.y)[0] = (uint16_t)pf_word;
  }
#endif
}

// Grouped launches (and 2-4 tokens): layer = blockIdx.y, all of its arguments in one batch of
// scalar loads (k256.h).
template <typename DT, int NS, int NST, bool PERM, bool FAST, int TOK>
__global__ __launch_bounds__(kMThreads) void gemv_k256m_kernel(const K256Params P) {
  int tokens;
  const K256Layer Ly = load_layer_args(tokens);
  gemv_k256m_body<DT, NS, NST, PERM, FAST, TOK>(Ly, tokens);
}

// One layer, one token: what the workgroup needs before its first vector loads comes as 14
// dwords of scalar kernel arguments, which are PRELOADED into SGPRs at wave launch
// (-mllvm -amdgpu-kernarg-preload-count=16, Makefile; 16 user SGPRs minus the segment pointer):
// the kernel starts without the round trip to the kernel-argument segment (~0.3 us per launch).
// The rest of the layer is fetched by ordinary scalar loads, waited for where first used.
template <typename DT, int NS, int NST, bool PERM, bool FAST>
__global__ __launch_bounds__(kMThreads) void gemv_k256m_kernel_1(
    const uint32_t* h_cent, const uint32_t* h_rcent, const uint16_t* h_x, const uint16_t* h_scale,
    int h_N, int h_G, int h_O, int h_row_words, int h_wgs, int h_slots, const K256Params P) {
  K256Layer Ly = P.layer[0];
  Ly.cent = h_cent; Ly.rcent = h_rcent; Ly.x = h_x; Ly.scale = h_scale;
  Ly.N = h_N; Ly.G = h_G; Ly.O = h_O; Ly.row_words = h_row_words;
  Ly.wgs = h_wgs; Ly.slots = h_slots;
  Ly.idx = as_global(Ly.idx); Ly.cent = as_global(Ly.cent); Ly.rcent = as_global(Ly.rcent);
  Ly.x = as_global(Ly.x); Ly.y = as_global(Ly.y); Ly.scale = as_global(Ly.scale);
  Ly.wbias = as_global(Ly.wbias); Ly.bias = as_global(Ly.bias); Ly.perm = as_global(Ly.perm);
  Ly.pf = as_global(Ly.pf);
  gemv_k256m_body<DT, NS, NST, PERM, FAST, 1>(Ly, P.tokens);
}

// ---- host side -------------------------------------------------------------------
static int device_cus() {
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

// LDS bytes before the partial-sum slots: image + per token the staged activations (0 columns:
// unstaged) and the per-wave sum b * x + slot counters
// sb: scale and bias planes behind the activations (bf16 exact form)
static int lds_fixed_bytes(int staged_cols, int tok, bool sb) {
  const int xb = tok == 1 ? kMXBytes1 : 2;
  return kMTableBytes + (staged_cols > 0 ? (tok + (sb ? 2 : 0)) * (staged_cols * xb + 32) : 0) + tok * kMWaves * 4 + 64 + (tok == 1 && !sb ? kMSelBytes : 0);   // (sb: no selective form, no corrections)
}

template <typename DT, int NS, int NST, bool PERM, bool FAST, int TOK>
static hipError_t launch_m(const K256Params& P, int gx, int max_cols, hipStream_t st) {
  const int fixed = lds_fixed_bytes(NST > 0 ? max_cols : 0, TOK, !FAST && (std::is_same<DT, BF16>::value || NS > 5 || TOK > 1 || VPTQ_K256M_SB_ALL));
  const int slots = P.layer[0].slots & 0xff;  // set by launch_gemv_k256m (bit 8: selective roundings)
  if (slots < 1 || slots > kMMaxSlots) return hipErrorInvalidValue;
  const int lds = fixed + slots * kMRedSlot * TOK;
  if (lds > kMMaxLds) return hipErrorInvalidValue;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  auto allow_lds = [&](const void* kern, std::atomic<bool>& done) {
    if (done) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, kMMaxLds);
    done = e == hipSuccess;
    return e;
  };
  if constexpr (TOK == 1) {
    if (P.n_layers == 1) {  // the preloaded-argument entry point
      auto kern = gemv_k256m_kernel_1<DT, NS, NST, PERM, FAST>;
      static std::atomic<bool> attr_set[64];
      if (hipError_t e = allow_lds((const void*)kern, attr_set[dev]); e != hipSuccess) return e;
      const K256Layer& L0 = P.layer[0];
      hipLaunchKernelGGL(kern, dim3(gx, 1), dim3(kMThreads), lds, st, L0.cent, L0.rcent, L0.x, L0.scale,
                         L0.N, L0.G, L0.O, L0.row_words, L0.wgs, L0.slots, P);
      return hipGetLastError();
    }
  }
  auto kern = gemv_k256m_kernel<DT, NS, NST, PERM, FAST, TOK>;
  static std::atomic<bool> attr_set[64];
  if (hipError_t e = allow_lds((const void*)kern, attr_set[dev]); e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(gx, P.n_layers), dim3(kMThreads), lds, st, P);
  return hipGetLastError();
}

template <typename DT, bool FAST, int TOK>
static hipError_t launch_m_shape(const K256Params& P, int gx, bool perm, int max_cols,
                                 hipStream_t st) {
  if (max_cols > kMMaxCols) {
    // wider than the LDS can stage: column blocks of 2 sweeps, scale and x through the queue
    if constexpr (FAST && TOK == 1) {
      if (!perm) return launch_m<DT, 2, 0, false, true, 1>(P, gx, max_cols, st);
    }
    return hipErrorInvalidValue;
  }
  const int ns = (max_cols + kMSweepCols - 1) / kMSweepCols;
#define K256M_CASE(S, N)                                                               \
  if (ns == S) return perm ? launch_m<DT, S, N, true, FAST, TOK>(P, gx, max_cols, st)  \
                           : launch_m<DT, S, N, false, FAST, TOK>(P, gx, max_cols, st);
  K256M_CASE(1, 1) K256M_CASE(2, 1) K256M_CASE(3, 1) K256M_CASE(4, 1)
  K256M_CASE(5, 2) K256M_CASE(6, 2) K256M_CASE(7, 2)
#undef K256M_CASE
  return hipErrorInvalidValue;
}

// partial-sum slots that fit beside everything else (0: the shape does not fit at all)
static int lds_slots(int tok, int max_cols, bool sb) {
  const int left = kMMaxLds - lds_fixed_bytes(max_cols > kMMaxCols ? 0 : max_cols, tok, sb);
  const int slots = left < 0 ? 0 : left / (kMRedSlot * tok);
  return slots > kMMaxSlots ? kMMaxSlots : slots;
}

// The instantiations are spread over four translation units (Makefile: -DVPTQ_K256M_PART=1..4
// on this same file) so that they compile side by side; PART 0 = everything in one.
#ifndef VPTQ_K256M_PART
#define VPTQ_K256M_PART 0
#endif
#define K256M_PART(n) (VPTQ_K256M_PART == 0 || VPTQ_K256M_PART == (n))
hipError_t k256m_f16_fast(const K256Params& P, int gx, bool perm, int max_cols, hipStream_t st);
hipError_t k256m_f16_exact(const K256Params& P, int gx, bool perm, int max_cols, hipStream_t st);
hipError_t k256m_bf16(const K256Params& P, int gx, bool perm, int max_cols, hipStream_t st);
hipError_t k256m_f16_tokens(const K256Params& P, int tok, int gx, bool perm, int max_cols, hipStream_t st);
hipError_t k256m_bf16_tokens(const K256Params& P, int tok, int gx, bool perm, int max_cols, hipStream_t st);
hipError_t k256m_bf16_exact(const K256Params& P, int gx, bool perm, int max_cols, hipStream_t st);
hipError_t k256m_f16_tokens_exact(const K256Params& P, int tok, int gx, bool perm, int max_cols, hipStream_t st);
hipError_t k256m_bf16_tokens_exact(const K256Params& P, int tok, int gx, bool perm, int max_cols, hipStream_t st);

#if K256M_PART(2)
hipError_t k256m_f16_exact(const K256Params& P, int gx, bool perm, int max_cols, hipStream_t st) {
  return launch_m_shape<F16, false, 1>(P, gx, perm, max_cols, st);
}
hipError_t k256m_bf16(const K256Params& P, int gx, bool perm, int max_cols, hipStream_t st) {
  return launch_m_shape<BF16, true, 1>(P, gx, perm, max_cols, st);
}
#endif
#if K256M_PART(3)
hipError_t k256m_f16_tokens(const K256Params& P, int tok, int gx, bool perm, int max_cols, hipStream_t st) {
  return tok == 2 ? launch_m_shape<F16, true, 2>(P, gx, perm, max_cols, st)
                  : launch_m_shape<F16, true, 4>(P, gx, perm, max_cols, st);
}
#endif
#if K256M_PART(4)
hipError_t k256m_bf16_tokens(const K256Params& P, int tok, int gx, bool perm, int max_cols, hipStream_t st) {
  return tok == 2 ? launch_m_shape<BF16, true, 2>(P, gx, perm, max_cols, st)
                  : launch_m_shape<BF16, true, 4>(P, gx, perm, max_cols, st);
}
hipError_t k256m_bf16_exact(const K256Params& P, int gx, bool perm, int max_cols, hipStream_t st) {
  return launch_m_shape<BF16, false, 1>(P, gx, perm, max_cols, st);
}
#endif
#if K256M_PART(5)
hipError_t k256m_f16_tokens_exact(const K256Params& P, int tok, int gx, bool perm, int max_cols, hipStream_t st) {
  return tok == 2 ? launch_m_shape<F16, false, 2>(P, gx, perm, max_cols, st)
                  : launch_m_shape<F16, false, 4>(P, gx, perm, max_cols, st);
}
#endif
#if K256M_PART(6)
hipError_t k256m_bf16_tokens_exact(const K256Params& P, int tok, int gx, bool perm, int max_cols, hipStream_t st) {
  return tok == 2 ? launch_m_shape<BF16, false, 2>(P, gx, perm, max_cols, st)
                  : launch_m_shape<BF16, false, 4>(P, gx, perm, max_cols, st);
}
#endif
#if K256M_PART(1)
hipError_t k256m_f16_fast(const K256Params& P, int gx, bool perm, int max_cols, hipStream_t st) {
  return launch_m_shape<F16, true, 1>(P, gx, perm, max_cols, st);
}

// exact form with scale and bias staged in LDS (kSB in the kernel): bf16, and fp16 from 6 sweeps on
static bool sb_staged(bool f16, bool fast, int max_cols, int tok) {
  return !fast && (!f16 || max_cols > 5 * kMSweepCols || tok > 1 || VPTQ_K256M_SB_ALL);
}
// tok = token slots of the instantiation (1, 2 or 4)
bool gemv_k256m_supported(int tok, bool f16, bool fast, int max_cols, bool perm) {
  // (the fp16 exact form queues scale and bias with the index words up to 5 sweeps; 6 and 7 sweeps - and bf16 - stage them in LDS: kSB)
  // more columns than fit beside the image: unstaged variant (folded form, no permutation)
  if (max_cols > kMMaxCols && (!fast || perm || tok > 1)) return false;
  // several tokens: one staged copy of the activations per token slot (the reference's roundings: scale and bias staged as well)
  if (tok != 1 && !(tok == 2 || tok == 4)) return false;
  // ... in the reference's roundings (round 6): the 4-slot instantiations keep 32 accumulator registers and spill from 3 (fp16) /
  // 2 (bf16) sweeps on; two passes of the 2-slot kernel lose against the VALU kernel (tools/tokens_exact_check.py: 4096 x 14336
  // bf16 30.2 vs 26.7 us, 4096^2 17.8 vs 10.0)
  if (tok == 4 && !fast && max_cols > (f16 ? 2 : 1) * kMSweepCols) return false;
  if (lds_slots(tok, max_cols, sb_staged(f16, fast, max_cols, tok)) < 1) return false;
  return true;
}

int gemv_k256m_row_groups(int n_rows) { return (n_rows + kMRows - 1) / kMRows; }

// All layers of a grouped launch must have the same number of columns (checked by the
// caller, launch_gemv_k256).  Fills in K256Layer::wgs: one workgroup per CU, shared out
// between the layers in proportion to their row groups.
// Workgroups per layer of one launch: one workgroup per CU (the LDS holds one), shared out in proportion to the layers' row
// groups - and never more than `cus` in total: rounding every share up gave 3 x 86 = 258 workgroups for three equal layers
// (q / k / v of an MHA model), two of which waited for a CU to come free (round 6).  While the sum is too large, the layer that
// loses least - whose row groups per workgroup stay the lowest after giving one up - gives one up.
static void m_shares(const int* groups, int n, int cus, int* share) {
  long long total = 0;
  for (int i = 0; i < n; ++i) total += groups[i];
  long long sum = 0;
  for (int i = 0; i < n; ++i) {
    long long sh = total > cus ? ((long long)groups[i] * cus + total - 1) / total : groups[i];
    if (sh < 1) sh = 1;
    if (sh > groups[i]) sh = groups[i];
    share[i] = (int)sh;
    sum += sh;
  }
  while (sum > cus) {
    int best = -1, best_units = 0;
    for (int i = 0; i < n; ++i) {
      if (share[i] <= 1) continue;
      const int units = (groups[i] + share[i] - 2) / (share[i] - 1);   // row groups per workgroup with one workgroup less
      if (best < 0 || units < best_units || (units == best_units && share[i] > share[best])) { best = i; best_units = units; }
    }
    if (best < 0) break;   // (more layers than CUs: cannot happen with kMaxGroup layers)
    --share[best];
    --sum;
  }
}
static int m_cus() {
  static std::atomic<int> forced_wgs{-1};  // VPTQ_K256M_WGS: tuning override of the CU count
  if (forced_wgs < 0) { const char* e = vptq::tune_env("VPTQ_K256M_WGS"); forced_wgs = e ? atoi(e) : 0; }
  const int fw = forced_wgs.load();
  return fw > 0 ? fw : device_cus();
}
int gemv_k256m_launch_units(const int* n_rows, int n) {
  if (n < 1 || n > kMaxGroup) return 0;
  int groups[kMaxGroup], share[kMaxGroup], units = 0;
  for (int i = 0; i < n; ++i) groups[i] = gemv_k256m_row_groups(n_rows[i]);
  m_shares(groups, n, m_cus(), share);
  for (int i = 0; i < n; ++i) {
    const int u = (groups[i] + share[i] - 1) / share[i];
    units = u > units ? u : units;
  }
  return units;
}
// VPTQ_GEMV_SELECTIVE in this kernel: one token, fp16, staged activations, the folded instantiation; every workgroup's row groups
// must fit the corrections' LDS (kMSelRowGroups)
bool gemv_k256m_selective_ok(const int* n_rows, int n, bool f16, int tok, int max_cols, bool perm) {
  if (tok != 1 || max_cols > kMMaxCols || max_cols > kMSelMaxBlocks * 128 || !gemv_k256m_supported(1, f16, true, max_cols, perm)) return false;
  if (n > kMaxGroup) return false;
  int groups[kMaxGroup], share[kMaxGroup];
  for (int i = 0; i < n; ++i) groups[i] = gemv_k256m_row_groups(n_rows[i]);
  m_shares(groups, n, m_cus(), share);
  for (int i = 0; i < n; ++i)
    if ((groups[i] + share[i] - 1) / share[i] > kMSelRowGroups) return false;
  return true;
}

hipError_t launch_gemv_k256m(K256Params& P, int tok, bool f16, bool fast, int max_cols, bool perm,
                             hipStream_t st, bool selective) {
  if (!gemv_k256m_supported(tok, f16, fast, max_cols, perm)) return hipErrorInvalidValue;
  if (selective && !(fast && tok == 1 && max_cols <= kMMaxCols)) return hipErrorInvalidValue;
  if (P.n_layers > kMaxGroup) return hipErrorInvalidValue;
  int groups[kMaxGroup], share[kMaxGroup];
  for (int i = 0; i < P.n_layers; ++i) groups[i] = gemv_k256m_row_groups(P.layer[i].N);
  m_shares(groups, P.n_layers, m_cus(), share);
  int gx = 0;
  for (int i = 0; i < P.n_layers; ++i) {
    P.layer[i].wgs = share[i];
    P.layer[i].slots = lds_slots(tok, max_cols, sb_staged(f16, fast, max_cols, tok)) | (selective ? kMSelBit : 0);
    gx = share[i] > gx ? share[i] : gx;
  }
  if (tok != 1) {
    if (!fast) return f16 ? k256m_f16_tokens_exact(P, tok, gx, perm, max_cols, st) : k256m_bf16_tokens_exact(P, tok, gx, perm, max_cols, st);
    return f16 ? k256m_f16_tokens(P, tok, gx, perm, max_cols, st)
               : k256m_bf16_tokens(P, tok, gx, perm, max_cols, st);
  }
  if (!f16) return fast ? k256m_bf16(P, gx, perm, max_cols, st) : k256m_bf16_exact(P, gx, perm, max_cols, st);
  return fast ? k256m_f16_fast(P, gx, perm, max_cols, st) : k256m_f16_exact(P, gx, perm, max_cols, st);
}
#endif  // part 1

}  // namespace vptq
