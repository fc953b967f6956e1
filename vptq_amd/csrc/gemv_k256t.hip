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
constexpr int kTSlots = 2;                           // cross-wave partial-sum slots
constexpr uint32_t kTRedOff = kTXsOff + kTWaves * kTXsWave;
constexpr uint32_t kTRedBOff = kTRedOff + kTSlots * kTWaves * 32 * 4;
constexpr uint32_t kTCntOff = kTRedBOff + kTSlots * kTWaves * 4;
constexpr uint32_t kTLdsBytes = kTCntOff + 64;

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
// timing-only ablations (results wrong): bit 0 no MFMAs, bit 1 no gathers
#ifndef VPTQ_K256T_ABLATE
#define VPTQ_K256T_ABLATE 0
#endif
#ifndef VPTQ_K256T_SPIN_LIMIT
#define VPTQ_K256T_SPIN_LIMIT 0
#endif

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

// position in the workgroup's flat stream of sweeps; everything is wave-uniform
struct TCursor {
  int L;    // layer; n_layers = past the end
  int rg;   // row group (4 vector-rows)
  int s;    // sweep inside the row group
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

  // ---- LDS map: [0, 64 Ki) image buffer 0 | [64 Ki, 128 Ki) image buffer 1 | per wave 256 B of
  // staged activations | partial-sum slots | counters
  const uint32_t xs_base = kTXsOff + (uint32_t)wave * kTXsWave;
  // A operand of tile pair p: 8 halves = f16(s x) of columns {8 (4 kg + e) + 2p + t}, t-major
  const uint32_t xa_addr = xs_base + (kg << 4);
  // staging: this lane loads columns 2 lane, 2 lane + 1 of the wave's block = chunk j = lane >> 2,
  // tiles 2 (lane & 3) and + 1 -> pair p = lane & 3, group j >> 2, e = j & 3
  const uint32_t st_addr = xs_base + (((uint32_t)lane & 3u) << 6) + (((uint32_t)lane >> 4) << 4) +
                           ((((uint32_t)lane >> 2) & 3u) << 1);
  float* const red = (float*)(smem + kTRedOff);      // [slot][wave][32]
  float* const red_b = (float*)(smem + kTRedBOff);   // [slot][wave]
  uint32_t* const slot_cnt = (uint32_t*)(smem + kTCntOff);  // [2] waves arrived
  uint32_t* const slot_done = slot_cnt + 2;                 // [2] row groups finished
  uint32_t* const free_cnt = slot_cnt + 4;                  // [2] waves that left a layer of image buffer b
  uint32_t* const ready_cnt = slot_cnt + 6;                 // [2] waves whose part of a fill has landed

  // ---- the flat stream ----
  // first layer >= c.L in which this workgroup owns a row group: its row groups are r0, r0 + W, ...
  // with r0 = (bid - first workgroup of the layer) mod W (K256Layer::wgs = the running total of
  // row groups mod W: the layers continue each other's round robin).  Returns that layer's
  // arguments (undefined past the end).
  auto enter_layer = [&](TCursor& c) -> K256Layer {
    K256Layer Ly = t_load_layer(c.L < n_layers ? c.L : n_layers - 1);
    while (c.L < n_layers) {
      c.ng = (Ly.N + 3) >> 2;
      c.ns = (Ly.G + kTSweepCols - 1) / kTSweepCols;
      int r0 = bid - Ly.wgs;
      if (r0 < 0) r0 += W;
      if (r0 < c.ng) { c.rg = r0; c.s = 0; break; }
      if (++c.L < n_layers) Ly = t_load_layer(c.L);
    }
    return Ly;
  };

  TCursor ci{0, 0, 0, 1, 1};   // issue position: always a valid sweep (it stops on the last one)
  bool ci_end = false;
  TIssueL Li;
  {
    const K256Layer L0 = enter_layer(ci);
    if (ci.L >= n_layers) return;     // (whole workgroup: nothing to do)
    Li = t_issue_of(L0);
  }
  TCursor cc = ci;             // consume position
  TConsL Lc;
  TFillL Lf;
  {
    const K256Layer L0 = t_load_layer(cc.L);
    Lc = t_cons_of(L0);
    Lf = TFillL{L0.cent, L0.rcent};
  }
  auto advance_issue = [&]() {
    if (ci_end) return;
    TCursor n = ci;
    if (++n.s < n.ns) { ci = n; return; }
    n.s = 0;
    n.rg += W;
    if (n.rg < n.ng) { ci = n; return; }
    ++n.L;
    const K256Layer Ln = enter_layer(n);
    if (n.L < n_layers) { ci = n; Li = t_issue_of(Ln); }
    else ci_end = true;   // past the end: harmless re-reads keep every step's set of loads the same
  };

  if (tid < 8) slot_cnt[tid] = 0u;
  __syncthreads();   // the only barrier: counters zeroed before anybody bumps them

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
  auto lds_inc = [&](uint32_t* p) {
    if (lane == 0) __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  };
  // VPTQ_K256T_SPIN_LIMIT (bring-up builds only): give up a wait after that many polls, so that a
  // protocol error shows as wrong results instead of a hung GPU
  auto lds_wait_ge = [&](uint32_t* p, uint32_t need) {
#if VPTQ_K256T_SPIN_LIMIT
    for (int it = 0; it < VPTQ_K256T_SPIN_LIMIT; ++it) {
      if (__hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) >= need) break;
      __builtin_amdgcn_s_sleep(1);
    }
#else
    while (__hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < need)
      __builtin_amdgcn_s_sleep(1);
#endif
  };

  // ---- loads of one sweep: 16 bytes of index words (8 columns of this lane's vector-row), and
  // for columns 2 lane, 2 lane + 1 of the wave's block: x, scale, bias.  Rows past N re-read
  // the last row (not stored), columns past G re-read the last ones with x forced to 0.
  u32x4 iw[2];
  uint32_t xr[2], sr[2], br[2];
  int q_layer[2] = {0, 0};    // DEP: layer and sweep of the item in each queue slot
  int q_sweep[2] = {0, 0};
  const uint32_t lane_chunk = (kg * 4u + src_e) * 8u;   // first of this lane's 8 columns inside the block
  auto load_x = [&](auto slot_c, const uint16_t* xp, int G, int s) {
    constexpr int S = decltype(slot_c)::value;
    const int want2 = s * kTSweepCols + wave * kTBlockCols + 2 * lane;
    xr[S] = *(const uint32_t*)as_global((const char*)xp + (uint32_t)(want2 < G ? want2 : G - 2) * 2u);
  };
  auto issue = [&](auto slot_c) {
    constexpr int S = decltype(slot_c)::value;
    const int colbase = ci.s * kTSweepCols + wave * kTBlockCols;
    const int row0 = ci.rg * 4;
    const uint32_t row_bytes = (uint32_t)Li.row_words * 4u;
    const char* const rbase = (const char*)Li.idx + (size_t)row0 * row_bytes;  // wave-uniform
    const uint32_t roff = (uint32_t)(row0 + (int)src_c < Li.N ? (int)src_c : Li.N - 1 - row0) * row_bytes;
    const int want = colbase + (int)lane_chunk;
    const uint32_t coff = (uint32_t)(want < Li.G ? want : Li.G - 8) * 2u;
    const int want2 = colbase + 2 * lane;
    const uint32_t c2 = (uint32_t)(want2 < Li.G ? want2 : Li.G - 2) * 2u;
    load_x(slot_c, Li.x, Li.G, ci.s);
    sr[S] = *(const uint32_t*)as_global((const char*)Li.scale + c2);
    br[S] = *(const uint32_t*)as_global((const char*)Li.wbias + c2);
    iw[S] = __builtin_nontemporal_load((const u32x4*)as_global(rbase + roff + coff));
    if (DEP) { q_layer[S] = ci.L; q_sweep[S] = ci.s; }
  };

  // ---- state of the consume side
  f32x4 acc[2];
  float accb = 0.f;
  uint32_t use = 0;            // how many layers this workgroup has entered before the current one
  uint32_t q_done = 0;         // row groups this workgroup has finished
  bool fill_pending = false;   // the next layer's image has not been requested yet
  bool land_pending = false;   // requested, not yet known to have landed
  TCursor cf = cc;             // next layer with work (its image goes to buffer (use + 1) & 1)
  auto plan_fill = [&]() {
    cf = cc;
    cf.L = cc.L + 1;
    const K256Layer Ln = enter_layer(cf);
    Lf = TFillL{Ln.cent, Ln.rcent};
    fill_pending = cf.L < n_layers;
  };

  // reduce over the 16 waves and store: lanes 0-15 of a wave hold outputs (row c', chunk, m) in
  // acc[h][0] - the MFMA has already summed over the four lane groups.  Slots + arrival counters
  // instead of a barrier (see gemv_k256m.hip): the wave that arrives last sums and stores.
  auto finish = [&]() {
    const int rg = cc.rg;
    const uint32_t slot = q_done & 1u;
    if (q_done >= (uint32_t)kTSlots) lds_wait_ge(&slot_done[slot], q_done - (uint32_t)kTSlots + 1u);
    float* const rs = red + slot * (kTWaves * 32) + wave * 32;
    if (lane < 16) {
      rs[res_c * 8u + ((0u ^ hold_h) << 2) + res_m] = acc[0][0];
      rs[res_c * 8u + ((1u ^ hold_h) << 2) + res_m] = acc[1][0];
    }
    const float sb = wave_sum(accb);
    if (lane == 0) red_b[slot * kTWaves + wave] = sb;
    uint32_t arrived = 0;
    if (lane == 0)
      arrived = __hip_atomic_fetch_add(&slot_cnt[slot], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
    arrived = __builtin_amdgcn_readfirstlane(arrived);
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
        __hip_atomic_store(&slot_done[slot], q_done + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
    ++q_done;
  };

  // ---- one sweep of this wave: 8 tiles = 4 tile pairs; per pair 8 transposing reads (2 tiles x
  // {X, Y} x {chunk, other chunk}) and 4 MFMAs of K = 32 (the two tiles of the pair side by side)
  auto consume = [&](auto slot_c) {
    constexpr int S = decltype(slot_c)::value;
    const int colbase = cc.s * kTSweepCols + wave * kTBlockCols;
    // activations: f16(s x) of this lane's two columns into the wave's slot, sum b x
    {
      const uint32_t keep = colbase + 2 * lane < Lc.G ? 0xffffffffu : 0u;
      const uint32_t xv = xr[S] & keep;
      accb = DT::dot2(xv, br[S], accb);
      // (anchored here: left alone, the compiler sinks this towards its use in finish(), keeps the
      // loaded register alive across the loop edge and copies it there - behind a wait for the
      // loads the step has just issued)
      asm volatile("" : "+v"(accb));
      const uint32_t xs = DT::mul2(xv, sr[S]);
      *(lds_u16_t*)(uintptr_t)st_addr = (uint16_t)(xs & 0xffffu);
      *(lds_u16_t*)(uintptr_t)(st_addr + 8u) = (uint16_t)(xs >> 16);
    }
    const u32x4 words = iw[S];
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
    const int need = (P.layer[L - 1].N + 3) >> 2;
#if VPTQ_K256T_SPIN_LIMIT
    for (int it = 0; it < VPTQ_K256T_SPIN_LIMIT; ++it) {
      if ((int)__hip_atomic_load(&P.sync[L - 1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= need) break;
      __builtin_amdgcn_s_sleep(2);
    }
#else
    while ((int)__hip_atomic_load(&P.sync[L - 1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < need)
      __builtin_amdgcn_s_sleep(2);
#endif
    const uint16_t* const xp = as_global(P.layer[L].x);
    const int G = P.layer[L].G;
    if (q_layer[0] == L) load_x(std::integral_constant<int, 0>{}, xp, G, q_sweep[0]);
    if (q_layer[1] == L) load_x(std::integral_constant<int, 1>{}, xp, G, q_sweep[1]);
  };

  // ---- prologue: image of the first layer into buffer 0, first two sweeps requested
  fill_image(Lf, 0u);
  issue(std::integral_constant<int, 0>{});
  advance_issue();
  issue(std::integral_constant<int, 1>{});
  advance_issue();
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // the 4 fill instructions are older than the 8 loads above
  lds_inc(&ready_cnt[0]);
  plan_fill();
  acc[0] = f32x4{0.f, 0.f, 0.f, 0.f};
  acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (DEP) dep_enter(cc.L);
  lds_wait_ge(&ready_cnt[0], (uint32_t)kTWaves);

  // ---- main loop: one step = wait for sweep k, consume it, request sweep k + 2 into its queue
  // slot, then the rare events (row group done, layer done, image fill)
  auto step = [&](auto slot_c) {
    __builtin_amdgcn_sched_barrier(0);
    consume(slot_c);
    __builtin_amdgcn_sched_barrier(0);
    // a fill issued one step ago is older than the loads the wait above left in flight
    if (land_pending) {
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      lds_inc(&ready_cnt[(use + 1u) & 1u]);
      land_pending = false;
    }
    // the next layer's image, as soon as its buffer is free: every wave has left the layer before
    // the current one (BEFORE this step's loads: a younger invisible load would make the next
    // step's counted wait cover those too)
    if (fill_pending) {
      const uint32_t nb = (use + 1u) & 1u;
      const uint32_t need = (uint32_t)kTWaves * ((use + 1u) >> 1);
      if (__hip_atomic_load(&free_cnt[nb], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) >= need) {
        fill_image(Lf, nb);
        fill_pending = false;
        land_pending = true;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    issue(slot_c);
    advance_issue();
    __builtin_amdgcn_sched_barrier(0);
    // consume cursor: row group / layer boundaries.  (A step past the end - the second half of the
    // last loop iteration when the stream has an odd number of sweeps - has consumed a re-read
    // sweep into accumulators nobody looks at.  The loop has ONE exit, at its end: an exit between
    // the two steps becomes, after control-flow structurisation, an edge into the loop header on
    // which the queue slots are in the other order, and every counted wait of the first step
    // degrades to vmcnt(0).)
    if (cc.L >= n_layers) return;
    if (cc.s + 1 < cc.ns) { ++cc.s; return; }
    finish();
    acc[0] = f32x4{0.f, 0.f, 0.f, 0.f};
    acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    accb = 0.f;
    cc.s = 0;
    if (cc.rg + W < cc.ng) { cc.rg += W; return; }
    // leaving the layer: its image buffer is free once every wave has said so
    lds_inc(&free_cnt[use & 1u]);
    if (cf.L >= n_layers) { cc.L = n_layers; return; }
    const uint32_t nb = (use + 1u) & 1u;
    if (fill_pending) {   // (rare: the buffer was not free at any step boundary)
      lds_wait_ge(&free_cnt[nb], (uint32_t)kTWaves * ((use + 1u) >> 1));
      fill_image(Lf, nb);
      fill_pending = false;
      land_pending = true;
    }
    if (land_pending) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      lds_inc(&ready_cnt[nb]);
      land_pending = false;
    }
    ++use;
    baseX ^= 0x10000u;
    baseY ^= 0x10000u;
    cc = cf;
    Lc = t_cons_of(t_load_layer(cc.L));
    if (DEP) dep_enter(cc.L);
    lds_wait_ge(&ready_cnt[nb], (uint32_t)kTWaves * ((use >> 1) + 1u));
    plan_fill();
  };
  do {
    step(std::integral_constant<int, 0>{});
    step(std::integral_constant<int, 1>{});
  } while (cc.L < n_layers);
}

// ---- host side -------------------------------------------------------------------
bool gemv_k256t_eligible(const VptqLayerDesc& d, int tokens) {
  return tokens == 1 && d.perm == nullptr && gemv_k256_eligible(d, 1) &&
         (((uintptr_t)d.centroids | (uintptr_t)d.res_centroids) & 15) == 0;
}

int gemv_k256t_grid(const VptqLayerDesc* descs, int n, int cus) {
  long long total = 0;
  for (int i = 0; i < n; ++i) total += (descs[i].num_indices + 3) / 4;
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
  const int grid = gemv_k256t_grid(descs, n, cus);
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
    Ly.wgs = (int)(first % grid);
    Ly.pf_chunk = 0;
    Ly.pf_len = 0;
    Ly.slots = 0;
    // dependent chain: every layer starts at workgroup 0 (all of its row groups wait anyway)
    first = dependent ? 0 : first + (d.num_indices + 3) / 4;
  }
  const bool f16 = descs[0].dtype == VPTQ_DTYPE_F16;
  if (dependent) return f16 ? launch_t<F16, true>(P, grid, st) : launch_t<BF16, true>(P, grid, st);
  return f16 ? launch_t<F16, false>(P, grid, st) : launch_t<BF16, false>(P, grid, st);
}

}  // namespace vptq
