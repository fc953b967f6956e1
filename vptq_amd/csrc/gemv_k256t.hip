// Fused dequant + GEMV for the canonical "2-bit" VPTQ format (v = 8, 256 + 256 centroids):
// ONE persistent launch that walks a CHAIN of layers, with a transposing LDS gather that feeds
// the matrix pipe a real contraction.  Same contract as gemv_k256m.hip / gemv_k256.hip, same
// reference (csrc/kernels/quant_gemv.cuh:11-186 + the tmp.sum of csrc/quant_gemv.cu:203-235).
//
// Why a third kernel.  gemv_k256m.hip pays, per 8192^2 layer, ~3 us of launch boundary +
// prologue + epilogue around ~4 us of accumulation, and its accumulate loop needs 55-63 SIMD
// cycles per index-wave (4 x v_mfma_4x4x4 + 4 x v_perm_b32: two of the perms only build the
// "x' * e_j" operand that turns an MFMA into 64 x 4 FMAs).  Here
//  * the gather is ds_read_b64_tr_b16 (gfx950): inside a group of 16 lanes, source lane
//    4e + c supplies the address of one 8-byte chunk and result lane 4c' + m receives element m
//    of the chunks of source lanes 4e + c', e = 0..3.  Source lane (e, c) owns the index of
//    (vector-row c, column e) of a 4 x 4 tile, so result lane (c', m) ends up with output m of
//    vector-row c' for FOUR columns: exactly the B operand of a contraction over columns
//    (k = column, n = (vector-row, output)).  The A operand is f16(scale * x) of the same
//    columns, identical in every row m - no operand has to be built per index any more:
//    per index-wave 2 v_perm_b32 (addresses) + 2 v_xor_b32 + 4 transposing reads +
//    2 v_mfma_f32_16x16x32 (32 matrix-pipe cycles, the floor of this formulation: every
//    gathered half is multiplied exactly once), and the MFMA sums over the 4 lane groups, so
//    a wave's 32 partial outputs need no cross-lane reduction at all.
//  * conflict-free image: row e (256 B) = 8 replicas of main entry e + 8 replicas of residual
//    entry e (16-byte units, low chunk = outputs 0-3, high chunk = outputs 4-7).  In each of
//    the 4 reads of an index a lane fetches another (table, chunk) combination, rotated by two
//    lane bits, so the 32 lanes of a pass touch 32 different 8-byte units.  Which combination a
//    result lane holds in which read is a per-lane constant: both tables go to the same
//    accumulator, the two chunks to two accumulators that are told apart in the epilogue.
//  * a wave owns 128 consecutive columns of a sweep (16 waves = 2048 columns) for 4 vector-rows:
//    one 16-byte index load per lane and sweep (8 columns of one row), the activations of its
//    128 columns staged by the wave itself (wave-private LDS slot: no barrier anywhere).
//  * persistent over LAYERS: the workgroup's work is one flat stream of sweeps (layer, row
//    group, sweep); index words, activations, scales and bias values are requested two sweeps
//    ahead, across row-group and layer boundaries; the next layer's codebook image is filled
//    into the second image buffer by LDS-DMA (global_load_lds_dwordx4, no registers) while the
//    current layer streams; cross-wave sums go through LDS slots with arrival counters.  HBM
//    never idles between layers, and launch boundary, prologue and epilogue are paid once per
//    chain instead of once per layer.
//  * DEP = the chain is dependent (x of layer i + 1 is y of layer i): a device-scope arrival
//    counter per layer; index words and image of the next layer are still requested ahead.
#include <type_traits>

#include "common.h"
#include "kernels.h"
#include "k256.h"

namespace vptq {

constexpr int kTThreads = 1024;
constexpr int kTWaves = kTThreads / 64;
constexpr int kTBlockCols = 128;                     // columns of one wave per sweep
constexpr int kTSweepCols = kTWaves * kTBlockCols;   // 2048
constexpr uint32_t kTImgBytes = 65536;               // 256 rows x 16 units x 16 B
constexpr uint32_t kTXsOff = 2 * kTImgBytes;         // wave-private activation slots
constexpr uint32_t kTXsWave = 256;
constexpr int kTSlots = 4;                           // cross-wave partial-sum slots
constexpr uint32_t kTRedOff = kTXsOff + kTWaves * kTXsWave;
constexpr uint32_t kTRedBOff = kTRedOff + kTSlots * kTWaves * 32 * 4;
constexpr uint32_t kTCntOff = kTRedBOff + kTSlots * kTWaves * 4;
constexpr uint32_t kTTabOff = kTCntOff + 64;            // the launch's layer arguments, 128 B per layer
constexpr uint32_t kTLdsBytes = kTTabOff + kMaxGroup * 128;

// transposing-gather convention (tools/tr_probe.hip prints what the hardware does):
// 0: source lane 4e + c -> result lane 4c + m, element e (ck_tile's Quad16 encoding)
// 1: source lane e + 4c
#ifndef VPTQ_K256T_TRVAR
#define VPTQ_K256T_TRVAR 0
#endif
// 1: the second chunk of an entry is read at address ^ 8 (conflict free); 0: at + 8 for every
// lane (no v_xor, 2-way bank conflicts)
#ifndef VPTQ_K256T_ROTH
#define VPTQ_K256T_ROTH 1
#endif
// timing-only ablations (results wrong): bit 0 no MFMAs, bit 1 no gathers, bit 2 no x / scale / bias loads,
// bit 3 no waits for the image hand-over between the waves, bit 4 index words read as 1 KiB contiguous per wave
#ifndef VPTQ_K256T_ABLATE
#define VPTQ_K256T_ABLATE 0
#endif
// sweeps in flight per wave (16 bytes of index words + 12 bytes of x / scale / bias per lane each):
// bandwidth x latency under load is ~2.3 us x 21 GB/s per CU = 48 KiB = 3 sweeps of the 16 waves
#ifndef VPTQ_K256T_DEPTH
#define VPTQ_K256T_DEPTH 4
#endif
#ifndef VPTQ_K256T_SPIN_LIMIT
#define VPTQ_K256T_SPIN_LIMIT 0
#endif
// profiling build (tools/chain_prof.py): every wave adds up the shader-clock cycles it spends
// waiting for its index words, consuming a sweep, requesting the next one and in the rare paths,
// and stores them at P.sync[(workgroup * 16 + wave) * 8 ...] (non-dependent launches)
#ifndef VPTQ_K256T_PROF
#define VPTQ_K256T_PROF 0
#endif
// the sweep's arithmetic: 1 = gather with ds_read_b128, accumulate with v_mfma_f32_4x4x4 as 64 x 4 FMAs
// (the loop of gemv_k256m.hip); 0 = transposing gather + v_mfma_f32_16x16x32 (file comment).  In
// isolation (tools/ubench_loop_t.hip, 4 waves per SIMD, no HBM): 34.6 against 46.8 SIMD cycles per
// index-wave - both sit on the LDS (32 cycles per index-wave and SIMD for the 2 KiB of gathered
// entries) and the matrix pipe (32), the 4x4x4 form overlaps them better.
#ifndef VPTQ_K256T_LOOP
#define VPTQ_K256T_LOOP 1
#endif
#ifndef VPTQ_K256T_BALANCE
#define VPTQ_K256T_BALANCE 1
#endif
constexpr int kTDepth = VPTQ_K256T_DEPTH;            // queue slots = sweeps in flight
static_assert(kTDepth >= 2 && kTDepth <= 8, "queue depth");

struct K256TParams {
  int n_layers;
  int tokens;        // token count | kOutF32Bit
  uint32_t* sync;    // DEP: one arrival counter per layer (zeroed before the launch)
  K256Layer layer[kMaxGroup];
};

typedef _Float16 h8v_t __attribute__((ext_vector_type(8)));
typedef __bf16 b8v_t __attribute__((ext_vector_type(8)));

template <typename DT>
static __device__ __forceinline__ f32x4 mfma16(u32x4 a, u32x4 b, f32x4 c) {
  if constexpr (std::is_same<DT, F16>::value)
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8v_t, a), __builtin_bit_cast(h8v_t, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(b8v_t, a), __builtin_bit_cast(b8v_t, b), c, 0, 0, 0);
}

static __device__ __forceinline__ u32x2 lds_tr8(uint32_t byte_addr) {
  typedef __attribute__((address_space(3))) s4_t lds_s4_t;
  return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_t*)(uintptr_t)byte_addr));
}
typedef __attribute__((address_space(3))) uint32_t lds_u32_t;
typedef __attribute__((address_space(3))) uint16_t lds_u16_t;

// layer L of the kernel arguments in one batch of scalar loads (see k256.h:load_layer_args)
static __device__ __forceinline__ K256Layer t_load_layer(int L) {
  static_assert(sizeof(K256Layer) == 120 && offsetof(K256TParams, layer) == 16, "kernarg layout");
  typedef int i16_t __attribute__((ext_vector_type(16)));
  typedef int i8_t __attribute__((ext_vector_type(8)));
  typedef int i4_t __attribute__((ext_vector_type(4)));
  typedef int i2_t __attribute__((ext_vector_type(2)));
  const char __attribute__((address_space(4)))* lp =
      (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr() + 16 +
      (size_t)L * sizeof(K256Layer);
  i16_t a; i8_t b; i4_t c; i2_t d;
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile(
      "s_load_dwordx16 %0, %4, 0x0\n\t"
      "s_load_dwordx8 %1, %4, 0x40\n\t"
      "s_load_dwordx4 %2, %4, 0x60\n\t"
      "s_load_dwordx2 %3, %4, 0x70\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&s"(a), "=&s"(b), "=&s"(c), "=&s"(d)
      : "s"(lp)
      : "memory");
#else
  a = i16_t{}; b = i8_t{}; c = i4_t{}; d = i2_t{}; (void)lp;
#endif
  K256Layer Ly;
  __builtin_memcpy((char*)&Ly, &a, 64);
  __builtin_memcpy((char*)&Ly + 64, &b, 32);
  __builtin_memcpy((char*)&Ly + 96, &c, 16);
  __builtin_memcpy((char*)&Ly + 112, &d, 8);
  Ly.idx = as_global(Ly.idx); Ly.cent = as_global(Ly.cent); Ly.rcent = as_global(Ly.rcent);
  Ly.x = as_global(Ly.x); Ly.y = as_global(Ly.y); Ly.scale = as_global(Ly.scale);
  Ly.wbias = as_global(Ly.wbias); Ly.bias = as_global(Ly.bias);
  return Ly;
}

// ... and out of the copy in LDS (kTTabOff; filled once per workgroup): a trip to the kernel-argument
// segment costs ~0.4 us, and a wave makes three of them per layer it enters
static __device__ __forceinline__ K256Layer t_load_layer_lds(int L) {
  const uint32_t a = kTTabOff + (uint32_t)L * 128u;
  uint32_t w[32];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const u32x4 v = lds_load16(a + 16u * i);
#pragma unroll
    for (int j = 0; j < 4; ++j) w[4 * i + j] = (uint32_t)__builtin_amdgcn_readfirstlane((int)v[j]);
  }
  K256Layer Ly;
  __builtin_memcpy((char*)&Ly, w, sizeof(K256Layer));
  Ly.idx = as_global(Ly.idx); Ly.cent = as_global(Ly.cent); Ly.rcent = as_global(Ly.rcent);
  Ly.x = as_global(Ly.x); Ly.y = as_global(Ly.y); Ly.scale = as_global(Ly.scale);
  Ly.wbias = as_global(Ly.wbias); Ly.bias = as_global(Ly.bias);
  return Ly;
}

// position in the workgroup's flat stream of sweeps; everything is wave-uniform
struct TCursor {
  int L;    // layer; n_layers = past the end
  int rg;   // row group (4 vector-rows)
  int re;   // end of this workgroup's block of row groups in layer L
  int ns;   // sweeps per row group of layer L
  int ng;   // row groups of layer L
};
// the fields of a layer each side needs (the rest of a K256Layer dies right after the load)
struct TIssueL { const uint32_t* idx; const uint16_t* x; const uint16_t* scale; const uint16_t* wbias; int N, G, row_words; };
struct TConsL { uint16_t* y; const uint16_t* bias; int N, G, O; };
struct TFillL { const uint32_t* cent; const uint32_t* rcent; };
static __device__ __forceinline__ TIssueL t_issue_of(const K256Layer& L) {
  return TIssueL{L.idx, L.x, L.scale, L.wbias, L.N, L.G, L.row_words};
}
static __device__ __forceinline__ TConsL t_cons_of(const K256Layer& L) { return TConsL{L.y, L.bias, L.N, L.G, L.O}; }

template <typename DT, bool DEP>
__global__ __launch_bounds__(kTThreads) void gemv_k256t_kernel(const K256TParams P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  {
    typedef __attribute__((address_space(3))) unsigned char lds_u8_t;
    if ((uint32_t)(uintptr_t)(lds_u8_t*)smem != 0u) __builtin_trap();  // absolute LDS addressing
  }
  const int n_layers = P.n_layers;
  const bool out_f32 = (P.tokens & kOutF32Bit) != 0;
  const int W = (int)gridDim.x, bid = (int)blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t kg = (uint32_t)lane >> 4, s16 = (uint32_t)lane & 15u;
  // this lane as the owner of an index = gather source: (column chunk e, vector-row c)
  const uint32_t src_e = VPTQ_K256T_TRVAR == 0 ? s16 >> 2 : s16 & 3u;
  const uint32_t src_c = VPTQ_K256T_TRVAR == 0 ? s16 & 3u : s16 >> 2;
  // this lane as the holder of a gathered operand = result: (vector-row c', output m)
  const uint32_t res_c = s16 >> 2, res_m = s16 & 3u;

  // ---- gather addresses.  Image row e: unit u (16 B) = replica u & 7 of table u >> 3 (main,
  // residual), low 8 bytes = outputs 0-3.  Read "X" of a lane takes table rot_b = kg & 1,
  // read "Y" the other one; the first read of each takes chunk rot_h = c & 1, the second the
  // other one.  Replica = the remaining 3 bits of (e, c): the 32 lanes of a pass (two lane
  // groups) then cover all 32 (table, chunk, replica) units of a row.
  const uint32_t rot_b = kg & 1u;
  const uint32_t rot_h = VPTQ_K256T_ROTH ? (src_c & 1u) : 0u;
  const uint32_t rep = (src_e << 1) | (src_c >> 1);
  uint32_t baseX = (rot_b << 7) | (rep << 4) | (rot_h << 3);          // byte 2 = image buffer
  uint32_t baseY = ((rot_b ^ 1u) << 7) | (rep << 4) | (rot_h << 3);
  // address = {0, base.byte2, index byte, base.byte0}; dword q of the index words holds columns
  // 2q (bytes 0 = main, 1 = residual index) and 2q + 1 (bytes 2, 3)
  const uint32_t selX[2] = {0x0c020000u | ((4u + rot_b) << 8), 0x0c020000u | ((6u + rot_b) << 8)};
  const uint32_t selY[2] = {0x0c020000u | ((4u + (rot_b ^ 1u)) << 8), 0x0c020000u | ((6u + (rot_b ^ 1u)) << 8)};
  // what this lane HOLDS after a read: vector-row c' of its group; chunk of read h = h ^ (c' & 1)
  const uint32_t hold_h = VPTQ_K256T_ROTH ? (res_c & 1u) : 0u;

  // ---- (VPTQ_K256T_LOOP = 1) lane = (column chunk blk = lane >> 2, vector-row j = lane & 3); the two
  // gathers of an index are split across the lanes: in gather A the lanes with bit 3 clear fetch the
  // main entry (unit lane & 7), the others the residual entry (unit 8 + (lane & 7)); gather B is the
  // complement: every 16-lane group of a ds_read_b128 touches 16 different units (gemv_k256m.hip)
  const uint32_t jrow = (uint32_t)lane & 3u;
  const uint32_t hi8 = ((uint32_t)lane >> 3) & 1u;
  uint32_t baseA = ((hi8 << 3) | ((uint32_t)lane & 7u)) << 4;            // byte 2 = image buffer
  uint32_t baseB = (((hi8 ^ 1u) << 3) | ((uint32_t)lane & 7u)) << 4;
  const uint32_t selGA[2] = {0x0c020400u | (hi8 << 8), 0x0c020600u | (hi8 << 8)};
  const uint32_t selGB[2] = {0x0c020400u | ((hi8 ^ 1u) << 8), 0x0c020600u | ((hi8 ^ 1u) << 8)};
  // x operand of the 4x4x4 MFMA: x' * e_j as two packed pairs, cut out of a packed x' register
  const uint32_t selA[2] = {jrow == 0 ? 0x0c0c0504u : jrow == 1 ? 0x05040c0cu : 0x0c0c0c0cu,
                            jrow == 0 ? 0x0c0c0706u : jrow == 1 ? 0x07060c0cu : 0x0c0c0c0cu};
  const uint32_t selB[2] = {jrow == 2 ? 0x0c0c0504u : jrow == 3 ? 0x05040c0cu : 0x0c0c0c0cu,
                            jrow == 2 ? 0x0c0c0706u : jrow == 3 ? 0x07060c0cu : 0x0c0c0c0cu};

  // ---- LDS map: [0, 64 Ki) image buffer 0 | [64 Ki, 128 Ki) image buffer 1 | per wave 256 B of
  // staged activations | partial-sum slots | counters
  const uint32_t xs_base = kTXsOff + (uint32_t)wave * kTXsWave;
  // A operand of tile pair p: 8 halves = f16(s x) of columns {8 (4 kg + e) + 2p + t}, t-major
  const uint32_t xa_addr = xs_base + (kg << 4);
  // staging: this lane loads columns 2 lane, 2 lane + 1 of the wave's block = chunk j = lane >> 2,
  // tiles 2 (lane & 3) and + 1 -> pair p = lane & 3, group j >> 2, e = j & 3
  const uint32_t st_addr = xs_base + (((uint32_t)lane & 3u) << 6) + (((uint32_t)lane >> 4) << 4) +
                           ((((uint32_t)lane >> 2) & 3u) << 1);
  // (VPTQ_K256T_LOOP = 1: the slot in column order; lane (blk, j) reads the 16 bytes of its chunk)
  const uint32_t st_addr1 = xs_base + (uint32_t)lane * 4u;
  const uint32_t xq_addr = xs_base + ((uint32_t)lane >> 2) * 16u;
  float* const red = (float*)(smem + kTRedOff);      // [slot][wave][32]
  float* const red_b = (float*)(smem + kTRedBOff);   // [slot][wave]
  uint32_t* const slot_cnt = (uint32_t*)(smem + kTCntOff);  // [kTSlots] waves arrived
  uint32_t* const slot_done = slot_cnt + kTSlots;           // [kTSlots] row groups finished
  uint32_t* const free_cnt = slot_cnt + 2 * kTSlots;        // [2] waves that left a layer of image buffer b
  uint32_t* const ready_cnt = free_cnt + 2;                 // [2] waves whose part of a fill has landed
  uint32_t* const dep_seen = free_cnt + 4;                  // DEP: last layer whose producers wave 0 has seen arrive

  // ---- the flat stream ----
  // first layer >= c.L in which this workgroup owns row groups: a block of K256Layer::pf_chunk
  // consecutive ones, block number (bid - first workgroup of the layer) mod W (K256Layer::wgs = the
  // running total of blocks mod W: the layers continue each other's round robin, so that layers
  // that need fewer than W blocks run side by side on different workgroups).  Returns that layer's
  // arguments (undefined past the end).
  auto enter_layer = [&](TCursor& c, auto from_lds) -> K256Layer {
    constexpr bool kLds = decltype(from_lds)::value;
    const int L0 = c.L < n_layers ? c.L : n_layers - 1;
    K256Layer Ly = kLds ? t_load_layer_lds(L0) : t_load_layer(L0);
    while (c.L < n_layers) {
      c.ng = (Ly.N + 3) >> 2;
      c.ns = (Ly.G + kTSweepCols - 1) / kTSweepCols;
      int r0 = bid - Ly.wgs;
      if (r0 < 0) r0 += W;
      r0 *= Ly.pf_chunk;
      if (r0 < c.ng) { c.rg = r0; c.re = r0 + Ly.pf_chunk < c.ng ? r0 + Ly.pf_chunk : c.ng; break; }
      if (++c.L < n_layers) Ly = kLds ? t_load_layer_lds(c.L) : t_load_layer(c.L);
    }
    return Ly;
  };
  using from_args = std::integral_constant<bool, false>;
  using from_table = std::integral_constant<bool, true>;
  // the layer table into LDS: one dword per thread (kMaxGroup x 30 <= 1024)
  {
    static_assert(kMaxGroup * 30 <= kTThreads, "one dword of the layer table per thread");
    const int tl = tid >> 5, tw = tid & 31;
    if (tl < n_layers && tw < 30) {
      const uint32_t* const src = (const uint32_t*)as_global(
          (const char*)(uintptr_t)__builtin_amdgcn_kernarg_segment_ptr() + 16 + tl * 120 + tw * 4);
      *(lds_u32_t*)(uintptr_t)(kTTabOff + (uint32_t)tl * 128u + (uint32_t)tw * 4u) = *src;
    }
  }

  // Issue side (D sweeps ahead of the consume side): cursor + incremental state, so that a
  // step costs a handful of instructions and everything rare sits behind ONE branch.
  TCursor ci{0, 0, 0, 1, 1};   // row group being requested
  bool ci_end = false;         // the stream has ended: the last row group is re-requested (harmless
                               // re-reads keep every step's set of loads the same)
  TIssueL Li;
  {
    const K256Layer L0 = enter_layer(ci, from_args{});
    if (ci.L >= n_layers) return;     // (whole workgroup: nothing to do)
    Li = t_issue_of(L0);
  }
  TCursor cc = ci;             // row group being consumed
  TConsL Lc;
  TFillL Lf;
  {
    const K256Layer L0 = t_load_layer(cc.L);
    Lc = t_cons_of(L0);
    Lf = TFillL{L0.cent, L0.rcent};
  }
  const uint32_t lane_chunk2 = (kg * 4u + src_e) * 16u;   // byte offset of this lane's 8 index elements in a block
  uint32_t i_rowoff = 0;   // per lane: byte offset of its vector-row in the layer's index tensor
  int i_col2 = 0;          // byte offset (2 x column) of the wave's block in the sweep to request
  int i_left = 0;          // sweeps of the row group still to request
  int i_max8 = 0, i_max2 = 0;   // 2 (G - 8), 2 (G - 2): columns past G re-read the last ones
  auto issue_row_group = [&]() {
    const int row0 = ci.rg * 4;
    const int r = row0 + (int)src_c < Li.N ? row0 + (int)src_c : Li.N - 1;   // rows past N re-read the last row
    i_rowoff = (uint32_t)r * ((uint32_t)Li.row_words * 4u);
    i_col2 = wave * (kTBlockCols * 2);
    i_left = ci.ns;
    i_max8 = (Li.G - 8) * 2;
    i_max2 = (Li.G - 2) * 2;
  };
  issue_row_group();
  auto issue_next_row_group = [&]() {   // cold
    if (!ci_end) {
      TCursor n = ci;
      n.rg += 1;
      if (n.rg >= n.re) {
        ++n.L;
        const K256Layer Ln = enter_layer(n, from_table{});
        if (n.L < n_layers) { ci = n; Li = t_issue_of(Ln); }
        else ci_end = true;
      } else {
        ci = n;
      }
    }
    issue_row_group();
  };

  if (tid < 16) slot_cnt[tid] = 0u;
  __syncthreads();   // the only barrier: counters zeroed and the layer table in LDS before anybody uses them

  // ---- image fill by LDS-DMA: wave w brings rows 16 w .. 16 w + 15 (4 instructions of 4 rows;
  // lane l = unit l & 15 of row l >> 4: 16 bytes of entry (row) of table (unit >> 3)).  Issued as
  // inline assembly: a DMA the compiler can see makes it wait for ALL loads before the next LDS
  // read.  Invisible loads can only make the compiler's counted waits stricter, never looser
  // (vmcnt retires in order).
  auto fill_image = [&](const TFillL& F, uint32_t buf) {
    const char* tab = (s16 >> 3) ? (const char*)F.rcent : (const char*)F.cent;
    const uint64_t va = (uint64_t)(uintptr_t)tab + (uint64_t)(((uint32_t)wave * 16u + kg) * 16u);
    const uint32_t dst = buf * kTImgBytes + (uint32_t)wave * 16u * 256u;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint32_t d = dst + (uint32_t)i * 1024u;
      const uint64_t v = va + (uint64_t)(i * 64);
      uint32_t keep_m0;   // (M0 belongs to the compiler: saved and restored inside the statement)
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep_m0) : "v"(v), "s"(d) : "memory");
    }
  };
  // Hand-over between the waves goes through LDS only, and a wave's LDS operations execute in
  // order: the fences below are restricted to the local address space.  An ordinary workgroup-scope
  // release also waits for every global load in flight (s_waitcnt vmcnt(0)) - here that is the whole
  // queue of index words requested ahead, i.e. a full memory latency at every row-group end, layer
  // switch and image hand-over (measured: 40 % of a wave's cycles went there).
  auto lds_release = [&]() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local"); };
  auto lds_acquire = [&]() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local"); };
  auto lds_inc = [&](uint32_t* p) {
    lds_release();
    if (lane == 0) __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  };
  // VPTQ_K256T_SPIN_LIMIT (bring-up builds only): give up a wait after that many polls, so that a
  // protocol error shows as wrong results instead of a hung GPU
  auto lds_wait_ge = [&](uint32_t* p, uint32_t need) {
    if constexpr ((VPTQ_K256T_ABLATE & 8) != 0) { if (p == free_cnt || p == free_cnt + 1 || p == ready_cnt || p == ready_cnt + 1) return; }
#if VPTQ_K256T_SPIN_LIMIT
    for (int it = 0; it < VPTQ_K256T_SPIN_LIMIT; ++it) {
      if (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= need) break;
      __builtin_amdgcn_s_sleep(1);
    }
#else
    while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < need)
      __builtin_amdgcn_s_sleep(1);
#endif
    lds_acquire();
  };

  // ---- loads of one sweep: 16 bytes of index words (8 columns of this lane's vector-row), and
  // for columns 2 lane, 2 lane + 1 of the wave's block: x, scale, bias.
#if VPTQ_K256T_PROF
  unsigned long long pf_vm = 0, pf_tv = 0;
  auto now = [&](uint32_t dep) -> unsigned long long {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : "v"(dep) : "memory");
    return t;
  };
#endif
  constexpr int D = kTDepth;
  u32x4 iw[D];
  uint32_t xr[D], sr[D], br[D];
  int q_layer[D], q_col2[D];   // DEP: layer and block offset of the sweep in each queue slot
  auto load_x = [&](auto slot_c, const uint16_t* xp, int max2, int col2) {
    constexpr int S = decltype(slot_c)::value;
    const int want = col2 + 4 * lane;
    xr[S] = *(const uint32_t*)as_global((const char*)xp + (uint32_t)(want < max2 ? want : max2));
  };
  auto issue = [&](auto slot_c) {
    constexpr int S = decltype(slot_c)::value;
    const int want = i_col2 + (int)lane_chunk2;
    const uint32_t coff = (uint32_t)(want < i_max8 ? want : i_max8);
    const int want2 = i_col2 + 4 * lane;
    const uint32_t c2 = (uint32_t)(want2 < i_max2 ? want2 : i_max2);
    if constexpr ((VPTQ_K256T_ABLATE & 4) != 0) {
      xr[S] = c2; sr[S] = 0x3c003c00u; br[S] = c2 ^ 0x1234u;
      asm volatile("" : "+v"(xr[S]), "+v"(sr[S]), "+v"(br[S]));
    } else {
      xr[S] = *(const uint32_t*)as_global((const char*)Li.x + c2);
      sr[S] = *(const uint32_t*)as_global((const char*)Li.scale + c2);
      br[S] = *(const uint32_t*)as_global((const char*)Li.wbias + c2);
    }
    if constexpr ((VPTQ_K256T_ABLATE & 16) != 0) {
      // timing only: the same bytes of the row group, but 1 KiB contiguous per wave and 16 KiB per sweep
      const uint32_t rg_base = (uint32_t)(ci.rg * 4) * ((uint32_t)Li.row_words * 4u);
      const uint32_t off = (uint32_t)(i_col2 / (kTBlockCols * 2)) * 1024u + (uint32_t)lane * 16u;   // sweep * 16 KiB + wave KiB
      iw[S] = __builtin_nontemporal_load((const u32x4*)as_global((const char*)Li.idx + (rg_base + off)));
    } else {
      iw[S] = __builtin_nontemporal_load((const u32x4*)as_global((const char*)Li.idx + (i_rowoff + coff)));
    }
    if (DEP) { q_layer[S] = ci_end ? -1 : ci.L; q_col2[S] = i_col2; }
#if VPTQ_K256T_PROF
    if (pf_tv) pf_vm += now(coff) - pf_tv;   // (pf_tv: stamped by the step right before issue())
#endif
    i_col2 += kTSweepCols * 2;
    if (--i_left == 0) issue_next_row_group();
  };

  // ---- state of the consume side
  f32x4 acc[2];
  float accb = 0.f;
  int c_col = wave * kTBlockCols;   // first column of the wave's block in the sweep being consumed
  int c_left = cc.ns;               // sweeps of the row group still to consume
  bool done = false;
  uint32_t use = 0;            // how many layers this workgroup has entered before the current one
  uint32_t q_done = 0;         // row groups this workgroup has finished
  bool fill_pending = false;   // the next layer's image has not been requested yet
  int land_steps = 0;          // > 0: a fill was requested D - land_steps step ends ago
#if VPTQ_K256T_PROF
  unsigned long long pf_wait = 0, pf_cons = 0, pf_issue = 0, pf_cold = 0, pf_steps = 0, pf_t0 = 0;
  unsigned long long pf_fwait = 0, pf_fred = 0, pf_ffinal = 0, pf_l0 = 0, pf_l1 = 0, pf_l2 = 0;
#endif
  // "everything but the youngest n sweeps' loads has landed".  vmcnt retires in order: at the end
  // of the j-th step after a fill (each step requests kLPS loads) the fill is older than j + 1
  // sweeps; at j = D - 1 those are exactly the D sweeps in flight.
  constexpr int kLPS = (VPTQ_K256T_ABLATE & 4) ? 1 : 4;   // vector loads per sweep
  auto wait_all_but = [&](int steps) {
    switch (steps) {
      case 1: asm volatile("s_waitcnt vmcnt(%0)" :: "n"(kLPS * 1) : "memory"); break;
      case 2: asm volatile("s_waitcnt vmcnt(%0)" :: "n"(kLPS * 2) : "memory"); break;
      case 3: asm volatile("s_waitcnt vmcnt(%0)" :: "n"(kLPS * 3) : "memory"); break;
      case 4: asm volatile("s_waitcnt vmcnt(%0)" :: "n"(kLPS * 4) : "memory"); break;
      case 5: asm volatile("s_waitcnt vmcnt(%0)" :: "n"(kLPS * 5) : "memory"); break;
      case 6: asm volatile("s_waitcnt vmcnt(%0)" :: "n"(kLPS * 6) : "memory"); break;
      case 7: asm volatile("s_waitcnt vmcnt(%0)" :: "n"(kLPS * 7) : "memory"); break;
      case 8: asm volatile("s_waitcnt vmcnt(%0)" :: "n"(kLPS * 8) : "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
  };
  TCursor cf = cc;             // next layer with work (its image goes to buffer (use + 1) & 1)
  auto plan_fill = [&]() {
    cf = cc;
    cf.L = cc.L + 1;
    const K256Layer Ln = enter_layer(cf, from_table{});
    Lf = TFillL{Ln.cent, Ln.rcent};
    fill_pending = cf.L < n_layers;
  };

  // reduce over the 16 waves and store: lanes 0-15 of a wave hold outputs (row c', chunk, m) in
  // acc[h][0] - the MFMA has already summed over the four lane groups.  Slots + arrival counters
  // instead of a barrier (see gemv_k256m.hip): the wave that arrives last sums and stores.
  auto finish = [&]() {
    const int rg = cc.rg;
    const uint32_t slot = q_done % (uint32_t)kTSlots;
#if VPTQ_K256T_PROF
    const unsigned long long f0 = now(0);
#endif
    if (q_done >= (uint32_t)kTSlots) lds_wait_ge(&slot_done[slot], q_done - (uint32_t)kTSlots + 1u);
#if VPTQ_K256T_PROF
    const unsigned long long f1 = now(0);
    pf_fwait += f1 - f0;
#endif
    float* const rs = red + slot * (kTWaves * 32) + wave * 32;
    if constexpr (VPTQ_K256T_LOOP == 1) {
      // lane (blk, j) holds 8 partial outputs of vector-row j for its column chunk: sum over the 16
      // chunks of the wave.  Lane bits 5 and 4 by swap-and-add (halving the values carried), bits 3
      // and 2 by DPP row rotations: afterwards lane l holds outputs 4 bit5 + 2 bit4 + {0, 1} of row l & 3
      float v[8];
#pragma unroll
      for (int i = 0; i < 4; ++i) { v[i] = acc[0][i]; v[4 + i] = acc[1][i]; }
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
        const int o8 = ((lane >> 5) & 1) * 4 + ((lane >> 4) & 1) * 2;
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        *(f32x2*)&rs[jrow * 8u + (uint32_t)o8] = f32x2{v[0], v[1]};
      }
    } else if (lane < 16) {
      rs[res_c * 8u + ((0u ^ hold_h) << 2) + res_m] = acc[0][0];
      rs[res_c * 8u + ((1u ^ hold_h) << 2) + res_m] = acc[1][0];
    }
    const float sb = wave_sum(accb);
    if (lane == 0) red_b[slot * kTWaves + wave] = sb;
    uint32_t arrived = 0;
    lds_release();
    if (lane == 0)
      arrived = __hip_atomic_fetch_add(&slot_cnt[slot], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    arrived = __builtin_amdgcn_readfirstlane(arrived);
    lds_acquire();
#if VPTQ_K256T_BALANCE
    // The SIMDs serve their waves oldest first, so the waves of a workgroup drift apart until the
    // fast ones wait for the slow ones at every hand-over.  Issue priority for the next row group
    // by arrival order at this one: the early ones yield.
    if (arrived < 4u) __builtin_amdgcn_s_setprio(0);
    else if (arrived < 8u) __builtin_amdgcn_s_setprio(1);
    else if (arrived < 12u) __builtin_amdgcn_s_setprio(2);
    else __builtin_amdgcn_s_setprio(3);
#endif
#if VPTQ_K256T_PROF
    const unsigned long long f2 = now(arrived);
    pf_fred += f2 - f1;
#endif
    if (arrived == (uint32_t)kTWaves - 1u) {
      const int ln = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
      const int half = ln >> 5, ol = ln & 31;
      const int row = rg * 4 + (ol >> 3);
      const int o = row * 8 + (ol & 7);
      const bool store = ln < 32 && row < Lc.N && o < Lc.O;
      const float* const ps = red + slot * (kTWaves * 32) + (half * 8) * 32 + ol;
      const float s0 = (ps[0] + ps[32]) + (ps[64] + ps[96]);
      const float s1 = (ps[128] + ps[160]) + (ps[192] + ps[224]);
      const float sum = s0 + s1;
      const float bdot = row16_allsum(red_b[slot * kTWaves + (ln & 15)]);
      float bv = 0.f;
      if (store && Lc.bias) bv = DT::to_float(as_global(Lc.bias)[o]);
      auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(sum), __float_as_uint(sum), false, false);
      const float total = (__uint_as_float(r[0]) + __uint_as_float(r[1])) + bdot;
      if (store) {
        if (out_f32) ((float*)as_global(Lc.y))[o] = total + bv;
        else as_global(Lc.y)[o] = DT::from_float(total + bv);
      }
      if (DEP) {
        // publish: the stores above, then this row group's arrival (device scope)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        if (lane == 0) __hip_atomic_fetch_add(&P.sync[cc.L], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (lane == 0) {
        slot_cnt[slot] = 0u;
        lds_release();
        __hip_atomic_store(&slot_done[slot], q_done + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
#if VPTQ_K256T_PROF
    pf_ffinal += now(0) - f2;
#endif
    ++q_done;
  };

  // ---- one sweep of this wave: 8 tiles = 4 tile pairs; per pair 8 transposing reads (2 tiles x
  // {X, Y} x {chunk, other chunk}) and 4 MFMAs of K = 32 (the two tiles of the pair side by side)
  auto consume = [&](auto slot_c) {
    constexpr int S = decltype(slot_c)::value;
    // activations: f16(s x) of this lane's two columns into the wave's slot, sum b x
    {
      const uint32_t keep = c_col + 2 * lane < Lc.G ? 0xffffffffu : 0u;
      const uint32_t xv = xr[S] & keep;
      accb = DT::dot2(xv, br[S], accb);
      // (anchored here: left alone, the compiler sinks this towards its use in finish(), keeps the
      // loaded register alive across the loop edge and copies it there - behind a wait for the
      // loads the step has just issued)
      asm volatile("" : "+v"(accb));
      const uint32_t xs = DT::mul2(xv, sr[S]);
      if constexpr (VPTQ_K256T_LOOP == 1) {
        *(lds_u32_t*)(uintptr_t)st_addr1 = xs;
      } else {
        *(lds_u16_t*)(uintptr_t)st_addr = (uint16_t)(xs & 0xffffu);
        *(lds_u16_t*)(uintptr_t)(st_addr + 8u) = (uint16_t)(xs >> 16);
      }
    }
    const u32x4 words = iw[S];
    if constexpr (VPTQ_K256T_LOOP == 1) {
      // 8 indices per lane: 2 gathers each, two indices ahead of the arithmetic; per index 2 perms for
      // the addresses, 2 for the x operand, 4 MFMAs (main / residual entry x outputs 0-3 / 4-7)
      const u32x4 xq = lds_load16(xq_addr);
      u32x4 cv[3], rv[3];
      auto gather = [&](int u) {
        const uint32_t w = words[u >> 1];
        const uint32_t aC = __builtin_amdgcn_perm(w, baseA, selGA[u & 1]);
        const uint32_t aR = __builtin_amdgcn_perm(w, baseB, selGB[u & 1]);
        if constexpr ((VPTQ_K256T_ABLATE & 2) != 0) {
          asm volatile("" :: "v"(aC), "v"(aR));
          cv[u % 3] = words; rv[u % 3] = words;
        } else {
          cv[u % 3] = lds_load16(aC);
          rv[u % 3] = lds_load16(aR);
        }
      };
      gather(0);
      gather(1);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        __builtin_amdgcn_sched_barrier(0);
        if (u + 2 < 8) gather(u + 2);
        __builtin_amdgcn_sched_barrier(0);
        const u32x4 c = cv[u % 3], r = rv[u % 3];
        const u32x2 xo = u32x2{__builtin_amdgcn_perm(xq[u >> 1], 0u, selA[u & 1]),
                               __builtin_amdgcn_perm(xq[u >> 1], 0u, selB[u & 1])};
        if constexpr ((VPTQ_K256T_ABLATE & 1) != 0) {
          asm volatile("" :: "v"(c), "v"(r), "v"(xo));
        } else {
          acc[0] = DT::mfma4(xo, u32x2{c[0], c[1]}, acc[0]);
          acc[1] = DT::mfma4(xo, u32x2{c[2], c[3]}, acc[1]);
          acc[0] = DT::mfma4(xo, u32x2{r[0], r[1]}, acc[0]);
          acc[1] = DT::mfma4(xo, u32x2{r[2], r[3]}, acc[1]);
        }
      }
      return;
    }
    u32x2 g[2][2][2][2];   // [pair parity][tile of the pair][read X / Y][first / second chunk]
    u32x4 xa[2];
    auto gather_pair = [&](int p) {
      uint32_t w = words[p];
      asm volatile("" : "+v"(w));   // (addresses derived where they are used, not all up front)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const uint32_t aX = __builtin_amdgcn_perm(w, baseX, selX[t]);
        const uint32_t aY = __builtin_amdgcn_perm(w, baseY, selY[t]);
        if constexpr ((VPTQ_K256T_ABLATE & 2) != 0) {
          asm volatile("" :: "v"(aX), "v"(aY));
          g[p & 1][t][0][0] = u32x2{w, aX}; g[p & 1][t][0][1] = u32x2{aX, w};
          g[p & 1][t][1][0] = u32x2{w, aY}; g[p & 1][t][1][1] = u32x2{aY, w};
        } else if constexpr (VPTQ_K256T_ROTH) {
          g[p & 1][t][0][0] = lds_tr8(aX);
          g[p & 1][t][0][1] = lds_tr8(aX ^ 8u);
          g[p & 1][t][1][0] = lds_tr8(aY);
          g[p & 1][t][1][1] = lds_tr8(aY ^ 8u);
        } else {
          g[p & 1][t][0][0] = lds_tr8(aX);
          g[p & 1][t][0][1] = lds_tr8(aX + 8u);
          g[p & 1][t][1][0] = lds_tr8(aY);
          g[p & 1][t][1][1] = lds_tr8(aY + 8u);
        }
      }
      xa[p & 1] = lds_load16(xa_addr + (uint32_t)p * 64u);
    };
    gather_pair(0);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      __builtin_amdgcn_sched_barrier(0);
      if (p + 1 < 4) gather_pair(p + 1);
      __builtin_amdgcn_sched_barrier(0);
      const u32x4 a = xa[p & 1];
#pragma unroll
      for (int r = 0; r < 2; ++r) {      // X, Y
#pragma unroll
        for (int h = 0; h < 2; ++h) {    // first / second chunk -> accumulator h
          const u32x2 t0 = g[p & 1][0][r][h], t1 = g[p & 1][1][r][h];
          const u32x4 b = u32x4{t0[0], t0[1], t1[0], t1[1]};
          if constexpr ((VPTQ_K256T_ABLATE & 1) != 0) asm volatile("" :: "v"(a), "v"(b));
          else acc[h] = mfma16<DT>(a, b, acc[h]);
        }
      }
    }
  };

  // DEP: layer L > 0 reads what layer L - 1 of this launch wrote: wait until all of its row
  // groups have arrived, then fetch x for the sweeps of layer L that are already in the queue
  // (their index words, scales and bias values were requested ahead; x could not be)
  auto dep_enter = [&](int L) {
    if (L == 0) return;
    // one wave per workgroup polls the device-scope counter (4096 waves polling one line starve
    // the atomics that feed it); the others wait for its word in LDS
    if (wave == 0) {
      const int need = (P.layer[L - 1].N + 3) >> 2;
#if VPTQ_K256T_SPIN_LIMIT
      for (int it = 0; it < VPTQ_K256T_SPIN_LIMIT; ++it) {
        if ((int)__hip_atomic_load(&P.sync[L - 1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= need) break;
        __builtin_amdgcn_s_sleep(4);
      }
#else
      while ((int)__hip_atomic_load(&P.sync[L - 1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < need)
        __builtin_amdgcn_s_sleep(4);
#endif
      lds_release();
      if (lane == 0) __hip_atomic_store(dep_seen, (uint32_t)L, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else {
      lds_wait_ge(dep_seen, (uint32_t)L);
      // (this wave's own acquire: its L1 may hold stale lines of the activation buffer)
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    const uint16_t* const xp = as_global(P.layer[L].x);
    const int max2 = (P.layer[L].G - 2) * 2;
    auto reload = [&](auto slot_c) {
      constexpr int S = decltype(slot_c)::value;
      if (q_layer[S] == L) load_x(slot_c, xp, max2, q_col2[S]);
    };
    reload(std::integral_constant<int, 0>{});
    reload(std::integral_constant<int, 1>{});
    if constexpr (D > 2) reload(std::integral_constant<int, (D > 2 ? 2 : 0)>{});
    if constexpr (D > 3) reload(std::integral_constant<int, (D > 3 ? 3 : 0)>{});
    if constexpr (D > 4) reload(std::integral_constant<int, (D > 4 ? 4 : 0)>{});
    if constexpr (D > 5) reload(std::integral_constant<int, (D > 5 ? 5 : 0)>{});
    if constexpr (D > 6) reload(std::integral_constant<int, (D > 6 ? 6 : 0)>{});
    if constexpr (D > 7) reload(std::integral_constant<int, (D > 7 ? 7 : 0)>{});
  };

  // ---- prologue: image of the first layer into buffer 0, first D sweeps requested
  fill_image(Lf, 0u);
  auto issue_first = [&](auto slot_c) {
    issue(slot_c);
    __builtin_amdgcn_sched_barrier(0);   // (the slots in issue order: the counted waits of the loop rely on it)
  };
  issue_first(std::integral_constant<int, 0>{});
  issue_first(std::integral_constant<int, 1>{});
  if constexpr (D > 2) issue_first(std::integral_constant<int, (D > 2 ? 2 : 0)>{});
  if constexpr (D > 3) issue_first(std::integral_constant<int, (D > 3 ? 3 : 0)>{});
  if constexpr (D > 4) issue_first(std::integral_constant<int, (D > 4 ? 4 : 0)>{});
  if constexpr (D > 5) issue_first(std::integral_constant<int, (D > 5 ? 5 : 0)>{});
  if constexpr (D > 6) issue_first(std::integral_constant<int, (D > 6 ? 6 : 0)>{});
  if constexpr (D > 7) issue_first(std::integral_constant<int, (D > 7 ? 7 : 0)>{});
  wait_all_but(D);   // the 4 fill instructions are older than the loads above
  lds_inc(&ready_cnt[0]);
  plan_fill();
  acc[0] = f32x4{0.f, 0.f, 0.f, 0.f};
  acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (DEP) dep_enter(cc.L);
  lds_wait_ge(&ready_cnt[0], (uint32_t)kTWaves);

  // ---- rare events, each behind one branch of the step
  // the next layer's image: requested as soon as its buffer is free (every wave has left the layer
  // before the current one), BEFORE the step's loads (a younger invisible load would make the next
  // step's counted wait cover those too); D step ends later only the D sweeps in flight are younger
  // than it, so it has landed once everything older has - which a wave that keeps pace has waited for
  auto fill_events_before_issue = [&]() {
    if (fill_pending) {
      const uint32_t nb = (use + 1u) & 1u;
      const uint32_t need = (uint32_t)kTWaves * ((use + 1u) >> 1);
      if (__hip_atomic_load(&free_cnt[nb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= need) {
        fill_image(Lf, nb);
        fill_pending = false;
        land_steps = D;
      }
    }
  };
  auto fill_events_after_issue = [&]() {
    if (land_steps > 0 && --land_steps == 0) {
      wait_all_but(D);
      lds_inc(&ready_cnt[(use + 1u) & 1u]);
    }
  };
  // the row group is complete: sums, then the next row group, the next layer, or the end
  auto row_group_done = [&]() {
    finish();
    acc[0] = f32x4{0.f, 0.f, 0.f, 0.f};
    acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    accb = 0.f;
    c_col = wave * kTBlockCols;
    c_left = cc.ns;
    if (cc.rg + 1 < cc.re) { cc.rg += 1; return; }
#if VPTQ_K256T_PROF
    const unsigned long long l0 = now(0);
#endif
    // leaving the layer: its image buffer is free once every wave has said so
    lds_inc(&free_cnt[use & 1u]);
    if (cf.L >= n_layers) {
      // (the steps that remain in this loop iteration consume re-read sweeps into accumulators
      // nobody looks at.  The loop has ONE exit, at its end: an exit between two steps becomes,
      // after control-flow structurisation, an edge into the loop header on which the queue slots
      // are in another order, and every counted wait of the first step degrades to vmcnt(0).)
      done = true;
      c_left = 0x7fffffff;
      return;
    }
    const uint32_t nb = (use + 1u) & 1u;
    if (fill_pending) {   // (rare: the buffer was not free at any step boundary)
      lds_wait_ge(&free_cnt[nb], (uint32_t)kTWaves * ((use + 1u) >> 1));
      fill_image(Lf, nb);
      fill_pending = false;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      lds_inc(&ready_cnt[nb]);
    } else if (land_steps > 0) {
      // requested D - land_steps step ends ago: that many sweeps are younger than the fill
      wait_all_but(D - land_steps);
      lds_inc(&ready_cnt[nb]);
      land_steps = 0;
    }
    ++use;
    baseX ^= 0x10000u;
    baseY ^= 0x10000u;
    baseA ^= 0x10000u;
    baseB ^= 0x10000u;
    cc = cf;
    Lc = t_cons_of(t_load_layer_lds(cc.L));
    c_left = cc.ns;
    if (DEP) dep_enter(cc.L);
#if VPTQ_K256T_PROF
    const unsigned long long l1 = now(Lc.G);
    pf_l0 += l1 - l0;
#endif
    lds_wait_ge(&ready_cnt[nb], (uint32_t)kTWaves * ((use >> 1) + 1u));
#if VPTQ_K256T_PROF
    const unsigned long long l2 = now(0);
    pf_l1 += l2 - l1;
#endif
    plan_fill();
#if VPTQ_K256T_PROF
    pf_l2 += now(cf.L) - l2;
#endif
  };

#if VPTQ_K256T_PROF
  pf_t0 = now(0);
#endif
  // ---- main loop: one step = wait for sweep k, consume it, request sweep k + D into its queue slot
  auto step = [&](auto slot_c) {
    __builtin_amdgcn_sched_barrier(0);
#if VPTQ_K256T_PROF
    constexpr int SS = decltype(slot_c)::value;
    const unsigned long long ta = now(0);
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(kLPS * (D - 1)) : "memory");
    const unsigned long long tb = now(iw[SS][0]);
    pf_wait += tb - ta;
#endif
    consume(slot_c);
    __builtin_amdgcn_sched_barrier(0);
#if VPTQ_K256T_PROF
    const unsigned long long tc = now(__float_as_uint(acc[0][0] + acc[1][0]));
    pf_cons += tc - tb;
#endif
    if (fill_pending) fill_events_before_issue();
    __builtin_amdgcn_sched_barrier(0);
#if VPTQ_K256T_PROF
    pf_tv = now(0);
#endif
    issue(slot_c);
    __builtin_amdgcn_sched_barrier(0);
#if VPTQ_K256T_PROF
    const unsigned long long td = now(0);
    pf_issue += td - tc;
#endif
    if (land_steps > 0) fill_events_after_issue();
    c_col += kTSweepCols;
    if (--c_left == 0) row_group_done();
#if VPTQ_K256T_PROF
    pf_cold += now(0) - td;
    ++pf_steps;
#endif
  };
  do {
    step(std::integral_constant<int, 0>{});
    step(std::integral_constant<int, 1>{});
    if constexpr (D > 2) step(std::integral_constant<int, (D > 2 ? 2 : 0)>{});
    if constexpr (D > 3) step(std::integral_constant<int, (D > 3 ? 3 : 0)>{});
    if constexpr (D > 4) step(std::integral_constant<int, (D > 4 ? 4 : 0)>{});
    if constexpr (D > 5) step(std::integral_constant<int, (D > 5 ? 5 : 0)>{});
    if constexpr (D > 6) step(std::integral_constant<int, (D > 6 ? 6 : 0)>{});
    if constexpr (D > 7) step(std::integral_constant<int, (D > 7 ? 7 : 0)>{});
  } while (!done);
#if VPTQ_K256T_PROF
  if (!DEP && P.sync && lane == 0) {
    unsigned long long* o = (unsigned long long*)P.sync + ((size_t)bid * kTWaves + wave) * 8;
    o[0] = pf_wait | (pf_vm << 32); o[1] = pf_cons; o[2] = pf_issue; o[3] = pf_cold; o[4] = pf_steps; o[5] = now(0) - pf_t0;
    o[6] = (pf_fwait << 32) | (pf_l0 & 0xffffffffull); o[7] = (pf_fred << 32) | (pf_ffinal & 0xffffffffull);
    o[2] = (pf_issue & 0xffffffffull) | (pf_l1 << 32); o[3] = (pf_cold & 0xffffffffull) | (pf_l2 << 32);
  }
#endif
}

// ---- host side -------------------------------------------------------------------
bool gemv_k256t_eligible(const VptqLayerDesc& d, int tokens) {
  return tokens == 1 && d.perm == nullptr && gemv_k256_eligible(d, 1) &&
         (((uintptr_t)d.centroids | (uintptr_t)d.res_centroids) & 15) == 0;
}

// Row groups are dealt to the workgroups in blocks of consecutive ones, `rpw` per workgroup and
// layer.  Independent layers: at least kTMinSteps sweeps per visit of a layer (the next layer's
// image is requested at the start of the visit and lands D sweeps later; a layer that gives every
// workgroup one short row group would make all of them wait for it), so a small layer occupies
// only some of the workgroups and the next layers run beside it.  Dependent layers follow each
// other anyway: every layer is spread over all workgroups.
#ifndef VPTQ_K256T_MIN_STEPS
#define VPTQ_K256T_MIN_STEPS 8
#endif
constexpr int kTMinSteps = VPTQ_K256T_MIN_STEPS;
static int t_rows_per_wg(const VptqLayerDesc& d, int cus, bool dependent) {
  const int ng = (d.num_indices + 3) / 4, ns = (d.group_size + kTSweepCols - 1) / kTSweepCols;
  int rpw = (ng + cus - 1) / cus;
  if (!dependent) {
    const int want = (kTMinSteps + ns - 1) / ns;
    rpw = want > rpw ? want : rpw;
  }
  return rpw < 1 ? 1 : rpw;
}

int gemv_k256t_grid(const VptqLayerDesc* descs, int n, int cus, bool dependent) {
  long long total = 0;
  for (int i = 0; i < n; ++i) {
    const int ng = (descs[i].num_indices + 3) / 4, rpw = t_rows_per_wg(descs[i], cus, dependent);
    total += (ng + rpw - 1) / rpw;
  }
  return (int)(total < cus ? total : cus);
}

static int t_device_cus() {
  static int cus[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (!cus[dev]) {
    hipDeviceProp_t p;
    cus[dev] = hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0
                   ? p.multiProcessorCount : 256;
  }
  return cus[dev];
}

template <typename DT, bool DEP>
static hipError_t launch_t(const K256TParams& P, int grid, hipStream_t st) {
  auto kern = gemv_k256t_kernel<DT, DEP>;
  static bool attr_set[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (!attr_set[dev]) {
    const hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                             (int)kTLdsBytes);
    if (e != hipSuccess) return e;
    attr_set[dev] = true;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(kTThreads), kTLdsBytes, st, P);
  return hipGetLastError();
}

// n <= kMaxGroup layers, all gemv_k256t_eligible and of one dtype; sync = n counters (zeroed by
// the caller's memset node) when dependent
hipError_t launch_gemv_k256t(const VptqLayerDesc* descs, int n, const void* const* x, void* const* y,
                             int flags, bool dependent, uint32_t* sync, hipStream_t st) {
  if (n < 1 || n > kMaxGroup) return hipErrorInvalidValue;
  static int forced_wgs = -1;  // VPTQ_K256T_WGS: tuning override of the workgroup count
  if (forced_wgs < 0) { const char* e = getenv("VPTQ_K256T_WGS"); forced_wgs = e ? atoi(e) : 0; }
  const int cus = forced_wgs > 0 ? forced_wgs : t_device_cus();
  const int grid = gemv_k256t_grid(descs, n, cus, dependent);
  K256TParams P;
  P.n_layers = n;
  P.tokens = 1 | ((flags & VPTQ_GEMV_OUT_F32) ? kOutF32Bit : 0);
  P.sync = sync;
  long long first = 0;   // workgroup that owns row group 0 of the layer
  for (int i = 0; i < n; ++i) {
    const VptqLayerDesc& d = descs[i];
    K256Layer& Ly = P.layer[i];
    Ly.idx = (const uint32_t*)d.indices;
    Ly.cent = (const uint32_t*)d.centroids;
    Ly.rcent = (const uint32_t*)d.res_centroids;
    Ly.x = (const uint16_t*)x[i];
    Ly.y = (uint16_t*)y[i];
    Ly.scale = (const uint16_t*)d.weight_scale;
    Ly.wbias = (const uint16_t*)d.weight_bias;
    Ly.bias = (const uint16_t*)d.bias;
    Ly.perm = nullptr;
    Ly.pf = nullptr;
    Ly.pf_bytes = 0;
    Ly.N = d.num_indices;
    Ly.G = d.group_size;
    Ly.O = d.out_features;
    Ly.row_words = d.row_words;
    const int rpw = t_rows_per_wg(d, cus, dependent);
    Ly.wgs = (int)(first % grid);
    Ly.pf_chunk = rpw;   // (this kernel: row groups per workgroup)
    Ly.pf_len = 0;
    Ly.slots = 0;
    // dependent chain: every layer starts at workgroup 0 (all of its row groups wait anyway)
    first = dependent ? 0 : first + ((d.num_indices + 3) / 4 + rpw - 1) / rpw;
  }
  const bool f16 = descs[0].dtype == VPTQ_DTYPE_F16;
  if (dependent) return f16 ? launch_t<F16, true>(P, grid, st) : launch_t<BF16, true>(P, grid, st);
  return f16 ? launch_t<F16, false>(P, grid, st) : launch_t<BF16, false>(P, grid, st);
}

}  // namespace vptq
