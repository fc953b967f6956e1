// Batched decode for the canonical "2-bit" VPTQ format (v = 8, 256 + 256 centroids): 2 ... 16 tokens
// in ONE pass over the indices, at the cost of one token.  Same contract as gemm_k256.hip (the
// reference's quant_gemv for a handful of tokens, csrc/kernels/quant_gemv.cuh:11-186 + the tmp.sum
// of csrc/quant_gemv.cu:203-235; its v2 kernel takes < 16 tokens in one launch,
// csrc/quant_gemv_v2.cu:58), folded arithmetic of gemv_k256m.hip / gemv_k256c.hip:
//     y[m, o] = sum_col (c + r)[o] * f16(s_col x[m, col]) + sum_col b_col x[m, col] + bias[o]
//
// gemm_k256.hip dequantises a tile with the reference's roundings, writes it to LDS in operand
// order and multiplies: 18 VALU instructions per index and two barriers per 4096-index step made
// it 18 us per 8192^2 layer for ANY count in 5 ... 16 - a cliff after 4 tokens (10 us).  Here
//  * the gather is ds_read_b64_tr_b16 (gfx950): inside a group of 16 lanes, source lane 4e + c
//    supplies the address of an 8-byte chunk and result lane 4c' + m receives element m of the
//    chunks of source lanes 4e' + c', e' = 0..3.  Source lane (e, c) owns the index of (column e,
//    vector-row c) of a 4 x 4 tile, so result lane (c', m) ends up with output m of vector-row c'
//    for FOUR columns: the B operand of a contraction over columns (k = column, n = (vector-row,
//    output)) of v_mfma_f32_16x16x32 - nothing is dequantised, transposed or written anywhere.
//  * the M dimension of that MFMA is the TOKEN.  (gemv_k256c.hip measured the same loop with 16
//    identical A rows for one token: 46.8 SIMD cycles per index-wave against 34.6 for its 4x4x4
//    form - here the 16 rows are 16 tokens and come for free.)
//  * the A operand f16(s x) does not depend on the row group: a small pre-pass (gemm_k256t_prep)
//    writes it to the caller's workspace ONCE per call in MFMA-operand order (the 16 bytes a lane
//    supplies are contiguous; the permutation of a layer with `perm` is applied there too) together
//    with sum_col b_col x[m, col]; the main kernel loads operands with 16-byte loads straight into the
//    register queue, 3 sweeps ahead, next to the index words.  No activation staging in LDS.
//  * conflict-free codebook image as in gemv_k256c.hip (row e = 8 replicas of main entry e + 8 of
//    residual entry e), filled by LDS-DMA; in each of the 4 reads of an index a lane fetches another
//    (table, 8-byte half) combination, rotated by two lane bits, so the 32 lanes of a pass touch 32
//    different 8-byte units.  Which combination a result lane holds in which read is a per-lane
//    constant: both tables go to the same accumulator, the two halves to two accumulators that the
//    epilogue tells apart.
//  * a wave owns 128 consecutive columns of a 2048-column sweep for 4 vector-rows; the 16 waves'
//    partial sums (16 tokens x 32 outputs each) meet in LDS once per row group.
#include <type_traits>

#include "common.h"
#include "kernels.h"
#include "k256.h"

namespace vptq {

constexpr int kBTThreads = 1024;
constexpr int kBTWaves = kBTThreads / 64;
constexpr int kBTBlockCols = 128;                      // columns of one wave per sweep
constexpr int kBTSweepCols = kBTWaves * kBTBlockCols;  // 2048
constexpr int kBTRows = 4;                             // vector-rows per row group
constexpr int kBTTokens = 16;                          // token slots = rows of the MFMA
// sweeps in flight per wave.  2: a row group of 8192 columns is 4 sweeps = two rounds of the unrolled loop exactly;
// with 3 the loop ran 6 steps for 4 sweeps and requested 15 loads ahead: 19.9 against 15.6 us per 8192^2 layer at 16
// tokens, same box (profiles/r03/tokens_k256t_depth_ab.txt); 4 spills registers (25.6 us)
#ifndef VPTQ_K256BT_DEPTH
#define VPTQ_K256BT_DEPTH 2
#endif
constexpr int kBTDepth = VPTQ_K256BT_DEPTH;
static_assert(kBTDepth >= 2 && kBTDepth <= 4, "queue depth");
constexpr uint32_t kBTImgBytes = 65536;                // 256 rows x 16 units x 16 B
constexpr uint32_t kBTRedOff = kBTImgBytes;            // [2][wave][token][32 outputs] floats
constexpr uint32_t kBTRedBuf = kBTWaves * kBTTokens * 32 * 4;
constexpr uint32_t kBTLds = kBTRedOff + 2 * kBTRedBuf; // 128 KiB
constexpr int kBTLoadsPerStep = 5;                     // 1 x index words + 4 x A operand
static_assert(kBTLoadsPerStep * kBTDepth <= 63, "vmcnt is a 6-bit counter");

// timing-only ablations (results wrong): bit 0 no MFMAs, bit 1 no gathers, bit 2 no A-operand loads
#ifndef VPTQ_K256BT_ABLATE
#define VPTQ_K256BT_ABLATE 0
#endif

struct GemmK256TParams {
  const uint32_t* idx;
  const uint32_t* cent;
  const uint32_t* rcent;
  const uint32_t* xp;      // workspace: [128-column block][4 KiB in operand order] (bt_operand_offset)
  const float* bdot;       // workspace: sum_col b_col x[m, col] per token
  void* y;
  const uint16_t* bias;
  int N, G, O, row_words;
  int tokens, out_f32;
  int n_groups;            // row groups of kBTRows vector-rows
  int n_sweeps;            // sweeps per row group
};

typedef _Float16 bt_h8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bt_b8_t __attribute__((ext_vector_type(8)));
template <typename DT>
static __device__ __forceinline__ f32x4 bt_mfma(u32x4 a, u32x4 b, f32x4 c) {
  if constexpr (std::is_same<DT, F16>::value)
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(bt_h8_t, a), __builtin_bit_cast(bt_h8_t, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bt_b8_t, a), __builtin_bit_cast(bt_b8_t, b), c, 0, 0, 0);
}
static __device__ __forceinline__ u32x2 bt_lds_tr8(uint32_t byte_addr) {
  typedef __attribute__((address_space(3))) s4_t lds_s4_t;
  return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_t*)(uintptr_t)byte_addr));
}

// Operand-ordered activations in the workspace: per 128-column block 4 KiB = [K-step p][lane group kg][token
// slot m (16)][t][e] halves, column = 8 (4 kg + e) + 2 p + t of the block: lane group kg of the MFMA covers the
// 8-column chunks 4 kg + e (e = 0..3 = the gather source lanes), K-step p takes columns 2p, 2p + 1 (= t) of every
// chunk, and the 16 bytes a lane supplies are t-major.  For one K-step lane l = 16 kg + m of a wave reads bytes
// [16 l, 16 l + 16) of ONE KiB (token-major rows made every load touch 16 cache lines: 4x the time; [m][kg]
// order inside the KiB still gave neighbouring lanes addresses 64 bytes apart).
static __host__ __device__ __forceinline__ uint32_t bt_operand_offset(uint32_t cb, uint32_t m) {
  const uint32_t chunk = cb >> 3, kg = chunk >> 2, e = chunk & 3u, p = (cb & 7u) >> 1, t = cb & 1u;
  return p * 1024u + kg * 256u + m * 16u + t * 8u + e * 2u;
}

// ---- pre-pass: one workgroup per (token, sweep of 2048 columns), a thread per 8-column chunk.  The 16 bytes of operand
// order (bt_operand_offset: [t][e] halves of K-step p, lane group kg) come from four chunks: the products go through
// LDS once (rows of 5 dwords: both the writes and the transposed reads are conflict-free) and leave as ONE 16-byte
// store per thread.  (First version: one workgroup per token and eight 2-byte stores per thread - 4.8 us of a 13.9 us
// call at 16 tokens.)  sum b x per (token, sweep); the main kernel adds the sweeps' parts in order.
constexpr int kBTPrepThreads = 256;   // = chunks of a sweep (16 blocks of 16)
template <typename DT>
__global__ __launch_bounds__(kBTPrepThreads) void gemm_k256t_prep(const uint16_t* __restrict__ x, const uint16_t* __restrict__ scale,
                                                                  const uint16_t* __restrict__ wbias, const uint16_t* __restrict__ perm,
                                                                  uint32_t* __restrict__ xp, float* __restrict__ bdot, int I, int G,
                                                                  int n_sweeps) {
  __shared__ uint32_t prod[kBTPrepThreads * 5];
  __shared__ float part[kBTPrepThreads / 64];
  const int m = blockIdx.x, sw = blockIdx.y, tid = threadIdx.x;
  const uint16_t* const xr = x + (size_t)m * I;
  const bool vec = perm == nullptr && (((uintptr_t)xr | (uintptr_t)scale | (uintptr_t)wbias) & 15) == 0;
  const int c0 = 8 * (sw * kBTPrepThreads + tid);
  u32x4 xv = {0u, 0u, 0u, 0u}, sv = {0u, 0u, 0u, 0u}, bv = {0u, 0u, 0u, 0u};   // columns past G: x' = 0
  if (vec && c0 + 8 <= G) {
    xv = *(const u32x4*)(xr + c0);
    sv = *(const u32x4*)(scale + c0);
    bv = *(const u32x4*)(wbias + c0);
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = c0 + j;
      if (c < G) {
        const int f = perm ? (int)perm[c] : c;   // column c of the quantised matrix multiplies input feature perm[c]
        xv[j >> 1] |= (uint32_t)xr[f] << (16 * (j & 1));
        sv[j >> 1] |= (uint32_t)scale[f] << (16 * (j & 1));
        bv[j >> 1] |= (uint32_t)wbias[f] << (16 * (j & 1));
      }
    }
  }
  float acc = 0.f;
#pragma unroll
  for (int pp = 0; pp < 4; ++pp) {
    acc = DT::dot2(xv[pp], bv[pp], acc);
    prod[tid * 5 + pp] = DT::mul2(xv[pp], sv[pp]);   // columns 2 pp (t = 0, low half), 2 pp + 1 (t = 1) of chunk tid
  }
  const float ws = wave_sum(acc);   // (fixed order: lanes of a wave, then the 4 waves)
  if ((tid & 63) == 0) part[tid >> 6] = ws;
  __syncthreads();
  {
    // thread = (block of the sweep, K-step p, lane group kg): halves t of chunks 4 kg + e, e = 0..3
    const int blk = tid >> 4, p = (tid >> 2) & 3, kg = tid & 3;
    const uint32_t* const w = prod + (blk * 16 + 4 * kg) * 5 + p;
    const uint32_t w0 = w[0], w1 = w[5], w2 = w[10], w3 = w[15];
    const u32x4 o = {__builtin_amdgcn_perm(w1, w0, 0x05040100u), __builtin_amdgcn_perm(w3, w2, 0x05040100u),
                     __builtin_amdgcn_perm(w1, w0, 0x07060302u), __builtin_amdgcn_perm(w3, w2, 0x07060302u)};
    *(u32x4*)(xp + ((size_t)(sw * 16 + blk) * 4096 + (size_t)p * 1024 + (size_t)kg * 256 + (size_t)m * 16) / 4) = o;
  }
  if (tid == 0) bdot[m * n_sweeps + sw] = (part[0] + part[1]) + (part[2] + part[3]);
}

// f(slot 0), ... f(slot kBTDepth - 1) with the slot as a compile-time constant
template <typename F>
static __device__ __forceinline__ void bt_for_slots(F&& f) {
  f(std::integral_constant<int, 0>{});
  f(std::integral_constant<int, 1>{});
  if constexpr (kBTDepth > 2) f(std::integral_constant<int, (kBTDepth > 2 ? 2 : 0)>{});
  if constexpr (kBTDepth > 3) f(std::integral_constant<int, (kBTDepth > 3 ? 3 : 0)>{});
}

template <typename DT>
__global__ __launch_bounds__(kBTThreads) void gemm_k256t_kernel(const GemmK256TParams P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  {
    typedef __attribute__((address_space(3))) unsigned char lds_u8_t;
    if ((uint32_t)(uintptr_t)(lds_u8_t*)smem != 0u) __builtin_trap();  // absolute LDS addressing
  }
  constexpr int D = kBTDepth;
  const int W = (int)gridDim.x, bid = (int)blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t kg = (uint32_t)lane >> 4, s16 = (uint32_t)lane & 15u;
  // this lane as the owner of an index = gather source: (column chunk e, vector-row c)
  const uint32_t src_e = s16 >> 2, src_c = s16 & 3u;
  // this lane as the holder of a gathered operand = result: (vector-row c', output m of the half)
  const uint32_t res_c = s16 >> 2, res_m = s16 & 3u;

  // ---- gather addresses.  Image row e: unit u (16 B) = replica u & 7 of table u >> 3 (main,
  // residual), low 8 bytes = outputs 0-3.  Read "X" of a lane takes table rot_b = kg & 1, read "Y"
  // the other one; the first read of each takes half rot_h = c & 1, the second the other one.
  // Replica = the remaining 3 bits of (e, c): the 32 lanes of a pass (two lane groups) then cover
  // all 32 (table, half, replica) units of a row.
  const uint32_t rot_b = kg & 1u;
  const uint32_t rot_h = src_c & 1u;
  const uint32_t rep = (src_e << 1) | (src_c >> 1);
  const uint32_t baseX = (rot_b << 7) | (rep << 4) | (rot_h << 3);
  const uint32_t baseY = ((rot_b ^ 1u) << 7) | (rep << 4) | (rot_h << 3);
  // address = {0, 0, index byte, base.byte0}; dword q of the index words holds columns 2q (byte 0 =
  // main, byte 1 = residual index) and 2q + 1 (bytes 2, 3)
  const uint32_t selX[2] = {0x0c0c0000u | ((4u + rot_b) << 8), 0x0c0c0000u | ((6u + rot_b) << 8)};
  const uint32_t selY[2] = {0x0c0c0000u | ((4u + (rot_b ^ 1u)) << 8), 0x0c0c0000u | ((6u + (rot_b ^ 1u)) << 8)};
  // what this lane HOLDS after a read: vector-row c' of its group; half of read h = h ^ (c' & 1)
  const uint32_t hold_h = res_c & 1u;

  const uint32_t* const idx = as_global(P.idx);
  const uint32_t* const xp = as_global(P.xp);
  const int ns = P.n_sweeps, n_groups = P.n_groups;

  // ---- image fill by LDS-DMA: wave w brings rows 16 w .. 16 w + 15 (gemv_k256c.hip:fill_image)
  {
    const uint32_t unit = (uint32_t)lane & 15u, r4 = (uint32_t)lane >> 4;
    const uint64_t tc = (uint64_t)(uintptr_t)as_global(P.cent), tr = (uint64_t)(uintptr_t)as_global(P.rcent);
    const uint64_t pick = (unit >> 3) ? ~0ull : 0ull;   // (by arithmetic: a select of two pointers becomes a scratch array)
    const uint64_t va = tc + ((tr - tc) & pick) + (uint64_t)(((uint32_t)wave * 16u + r4) * 16u);
    const uint32_t dst = (uint32_t)wave * 16u * 256u;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint32_t d = (uint32_t)__builtin_amdgcn_readfirstlane((int)(dst + (uint32_t)i * 1024u));
      const uint64_t v = va + (uint64_t)(i * 64);
      uint32_t keep_m0;   // (M0 belongs to the compiler: saved and restored inside the statement)
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep_m0) : "v"(v), "s"(d) : "memory");
    }
  }

  // ---- issue side: the workgroup's row groups bid, bid + W, ... as one flat stream of sweeps
  u32x4 iw[D];
  u32x4 xa[D][4];
  int i_rg = bid, i_s = 0;
  uint32_t i_rowoff = 0;
  const uint32_t lane_chunk2 = (kg * 4u + src_e) * 16u;   // byte offset of this lane's 8 index elements in a block
  int i_max8 = (P.G - 8) * 2;
  // A operand: lane (kg, m = s16) supplies token m; token slots past the token count re-read the last token
  // (their rows of the result are never stored, and a row of D depends on its own row of A only).  (Loads
  // under a lane predicate instead: the compiler can no longer count them and every wait of the loop becomes
  // "all but 2".)
  const uint32_t tok = s16 < (uint32_t)P.tokens ? s16 : (uint32_t)P.tokens - 1u;
  uint32_t a_lane = kg * 256u + tok * 16u;
  uint32_t a_blk = 4096u;   // bytes of the activation operand per block of columns (0 past the end of the stream)
  auto issue_row_group = [&]() __attribute__((always_inline)) {
    const int want = i_rg * kBTRows + (int)src_c;
    const int r = want < P.N ? want : P.N - 1;   // rows past N re-read the last row (not stored)
    i_rowoff = (uint32_t)r * ((uint32_t)P.row_words * 4u);
  };
  issue_row_group();
  auto issue = [&](auto slot_c) __attribute__((always_inline)) {
    constexpr int S = decltype(slot_c)::value;
    const int blk = i_s * kBTWaves + wave;
    const int want = blk * (kBTBlockCols * 2) + (int)lane_chunk2;
    const uint32_t coff = (uint32_t)(want < i_max8 ? want : i_max8);   // columns past G re-read the last ones (x' = 0 there)
    iw[S] = __builtin_nontemporal_load((const u32x4*)((const char*)idx + (i_rowoff + coff)));
    const char* const ap = (const char*)xp + (a_lane + (uint32_t)blk * a_blk);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      if constexpr ((VPTQ_K256BT_ABLATE & 4) != 0) {
        xa[S][p] = u32x4{coff, 0x3c003c00u, a_lane, 0x38003800u};
        asm volatile("" : "+v"(xa[S][p]));
      } else {
        xa[S][p] = *(const u32x4*)(ap + p * 1024);
      }
    }
    if (++i_s == ns) {
      i_s = 0;
      if (i_rg + W < n_groups) {
        i_rg += W;
        issue_row_group();
      } else {
        // past the end of the stream the steps still issue their loads (the waits stay counted): every lane reads
        // the first bytes of the tensors - cached lines - instead of re-reading the last row group and its operands
        i_rowoff = 0u; i_max8 = 0; a_lane = 0u; a_blk = 0u;
      }
    }
  };

  // ---- consume side
  f32x4 acc[2];   // [first / second half read]: rows = tokens 4 kg + i, column = (vector-row c', output m)
  acc[0] = f32x4{0.f, 0.f, 0.f, 0.f};
  acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
  int c_rg = bid, c_left = ns;
  bool done = false;
  uint32_t red_buf = 0;
  auto consume = [&](auto slot_c) __attribute__((always_inline)) {
    constexpr int S = decltype(slot_c)::value;
    const u32x4 words = iw[S];
    u32x2 g[2][2][2][2];   // [K-step parity][tile of the step][read X / Y][first / second half]
    auto gather_pair = [&](int p) __attribute__((always_inline)) {
      uint32_t w = words[p];
      asm volatile("" : "+v"(w));   // (addresses derived where they are used, not all up front)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const uint32_t aX = __builtin_amdgcn_perm(w, baseX, selX[t]);
        const uint32_t aY = __builtin_amdgcn_perm(w, baseY, selY[t]);
        if constexpr ((VPTQ_K256BT_ABLATE & 2) != 0) {
          asm volatile("" :: "v"(aX), "v"(aY));
          g[p & 1][t][0][0] = u32x2{w, aX}; g[p & 1][t][0][1] = u32x2{aX, w};
          g[p & 1][t][1][0] = u32x2{w, aY}; g[p & 1][t][1][1] = u32x2{aY, w};
        } else {
          g[p & 1][t][0][0] = bt_lds_tr8(aX);
          g[p & 1][t][0][1] = bt_lds_tr8(aX ^ 8u);
          g[p & 1][t][1][0] = bt_lds_tr8(aY);
          g[p & 1][t][1][1] = bt_lds_tr8(aY ^ 8u);
        }
      }
    };
    gather_pair(0);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      // fenced: left alone, the scheduler sinks the gathers next to their use (LDS latency exposed)
      __builtin_amdgcn_sched_barrier(0);
      if (p + 1 < 4) gather_pair(p + 1);
      __builtin_amdgcn_sched_barrier(0);
      const u32x4 a = xa[S][p];
#pragma unroll
      for (int r = 0; r < 2; ++r) {      // X, Y
#pragma unroll
        for (int h = 0; h < 2; ++h) {    // first / second half -> accumulator h
          const u32x2 t0 = g[p & 1][0][r][h], t1 = g[p & 1][1][r][h];
          const u32x4 b = u32x4{t0[0], t0[1], t1[0], t1[1]};
          if constexpr ((VPTQ_K256BT_ABLATE & 1) != 0) asm volatile("" :: "v"(a), "v"(b));
          else acc[h] = bt_mfma<DT>(a, b, acc[h]);
        }
      }
    }
  };

  // Waves meet at a raw s_barrier with fences restricted to the LOCAL address space: an ordinary
  // __syncthreads() also waits for every global load in flight - here the whole register queue.
  auto lds_barrier = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
  };

  // the row group is complete: the 16 waves' partial sums (their column blocks) meet in LDS; threads
  // 0-511 = (token, output of the row group) sum 16 partials each, add sum b x and the bias, store
  auto row_group_done = [&]() __attribute__((always_inline)) {
    float* const red = (float*)(smem + kBTRedOff + red_buf * kBTRedBuf);
    {
      float* const rs = red + wave * (kBTTokens * 32);
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          rs[(kg * 4u + (uint32_t)i) * 32u + res_c * 8u + ((((uint32_t)h) ^ hold_h) << 2) + res_m] = acc[h][i];
    }
    lds_barrier();
    if (tid < kBTTokens * 32) {
      const int m = tid >> 5, o32 = tid & 31;
      float s[4];
#pragma unroll
      for (int k = 0; k < 4; ++k)
        s[k] = (red[(4 * k) * 512 + tid] + red[(4 * k + 1) * 512 + tid]) + (red[(4 * k + 2) * 512 + tid] + red[(4 * k + 3) * 512 + tid]);
      const float sum = (s[0] + s[1]) + (s[2] + s[3]);
      const int row = c_rg * kBTRows + (o32 >> 3);
      const int o = row * 8 + (o32 & 7);
      if (m < P.tokens && row < P.N && o < P.O) {
        float bd = 0.f;   // sum b x of token m: the sweeps' parts in order
        for (int sw = 0; sw < P.n_sweeps; ++sw) bd += as_global(P.bdot)[m * P.n_sweeps + sw];
        float v = sum + bd;
        if (P.bias) v += DT::to_float(as_global(P.bias)[o]);
        if (P.out_f32) ((float*)as_global(P.y))[(size_t)m * P.O + o] = v;
        else ((uint16_t*)as_global(P.y))[(size_t)m * P.O + o] = DT::from_float(v);
      }
    }
    // (no second barrier: the next row group's partial sums go to the other buffer, and a wave can
    // only write THIS buffer again after the barrier of the next row group, i.e. after every
    // reader above has passed it)
    red_buf ^= 1u;
    acc[0] = f32x4{0.f, 0.f, 0.f, 0.f};
    acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    c_left = ns;
    if (c_rg + W < n_groups) { c_rg += W; return; }
    // (the steps that remain in this loop iteration consume re-read sweeps into accumulators nobody
    // looks at: the loop has ONE exit, at its end - gemv_k256c.hip)
    done = true;
    c_left = 0x7fffffff;
  };

  // ---- prologue: first D sweeps requested, then the image has landed (older than those loads)
  bt_for_slots([&](auto slot_c) {
    issue(slot_c);
    __builtin_amdgcn_sched_barrier(0);   // (the slots in issue order: the counted waits of the loop rely on it)
  });
  asm volatile("s_waitcnt vmcnt(%0)" :: "n"(kBTLoadsPerStep * D) : "memory");
  lds_barrier();

  auto step = [&](auto slot_c) __attribute__((always_inline)) {
    __builtin_amdgcn_sched_barrier(0);
    if (!done) consume(slot_c);   // (the steps after the last row group only keep the loads counted)
    __builtin_amdgcn_sched_barrier(0);
    issue(slot_c);
    __builtin_amdgcn_sched_barrier(0);
    if (--c_left == 0) row_group_done();
  };
  do {
    bt_for_slots(step);
  } while (!done);
}

// ---- host side -------------------------------------------------------------------
bool gemm_k256t_eligible(const VptqLayerDesc& d, int tokens, int flags) {
  if (flags & VPTQ_GEMV_EXACT) return false;        // folded arithmetic only (gemm_k256.hip has the reference's roundings)
  if (!gemv_k256_eligible(d, 1)) return false;      // canonical format, norm on, aligned
  if ((((uintptr_t)d.centroids | (uintptr_t)d.res_centroids) & 15) != 0) return false;
  if (d.group_size < 8 || (d.group_size & 7)) return false;
  return tokens >= 1 && tokens <= kBTTokens;
}

static int bt_blocks(const VptqLayerDesc& d) {
  return ((d.group_size + kBTSweepCols - 1) / kBTSweepCols) * kBTWaves;
}
// workspace of one call: operand-ordered activations of 16 tokens + the bias dots per (token, sweep)
size_t gemm_k256t_workspace_bytes(const VptqLayerDesc& d) {
  const size_t dots = ((size_t)kBTTokens * (bt_blocks(d) / kBTWaves) * sizeof(float) + 255) / 256 * 256;
  return (size_t)kBTTokens * bt_blocks(d) * 256 + dots;
}

template <typename DT>
static hipError_t launch_bt(const VptqLayerDesc& d, const GemmK256TParams& P, const void* x, void* ws, int grid,
                            hipStream_t st) {
  auto kern = gemm_k256t_kernel<DT>;
  static std::atomic<bool> attr_set[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (!attr_set[dev]) {
    const hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBTLds);
    if (e != hipSuccess) return e;
    attr_set[dev] = true;
  }
  hipLaunchKernelGGL(gemm_k256t_prep<DT>, dim3(P.tokens, P.n_sweeps), dim3(kBTPrepThreads), 0, st, (const uint16_t*)x,
                     (const uint16_t*)d.weight_scale, (const uint16_t*)d.weight_bias, (const uint16_t*)d.perm,
                     (uint32_t*)ws, (float*)P.bdot, d.in_features, d.group_size, P.n_sweeps);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(kBTThreads), kBTLds, st, P);
  return hipGetLastError();
}

// 1 ... 16 tokens; ws = gemm_k256t_workspace_bytes(d) bytes, 16-byte aligned
hipError_t launch_gemm_k256t(const VptqLayerDesc& d, const void* x, void* y, int tokens, bool out_f32, void* ws,
                             hipStream_t st) {
  if (tokens < 1 || tokens > kBTTokens || !ws || (((uintptr_t)ws) & 15)) return hipErrorInvalidValue;
  GemmK256TParams P = {};
  P.idx = (const uint32_t*)d.indices;
  P.cent = (const uint32_t*)d.centroids;
  P.rcent = (const uint32_t*)d.res_centroids;
  P.xp = (const uint32_t*)ws;
  P.bdot = (const float*)((const char*)ws + (size_t)kBTTokens * bt_blocks(d) * 256);
  P.y = y;
  P.bias = (const uint16_t*)d.bias;
  P.N = d.num_indices; P.G = d.group_size; P.O = d.out_features; P.row_words = d.row_words;
  P.tokens = tokens; P.out_f32 = out_f32 ? 1 : 0;
  P.n_groups = (d.num_indices + kBTRows - 1) / kBTRows;
  P.n_sweeps = (d.group_size + kBTSweepCols - 1) / kBTSweepCols;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  static std::atomic<int> cus[64];
  if (!cus[dev]) {
    hipDeviceProp_t p;
    cus[dev] = hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
  }
  const int ncu = cus[dev].load();
  const int grid = P.n_groups < ncu ? P.n_groups : ncu;
  return d.dtype == VPTQ_DTYPE_F16 ? launch_bt<F16>(d, P, x, ws, grid, st) : launch_bt<BF16>(d, P, x, ws, grid, st);
}

}  // namespace vptq
