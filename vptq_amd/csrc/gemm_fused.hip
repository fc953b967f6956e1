// Fused dequant -> LDS -> MFMA GEMM for the canonical "2-bit" VPTQ format (v = 8, 256 + 256
// centroids, norm on): the prefill / many-token path, y[M, O] = x[M, I] * W^T (+ bias), with W
// never written to memory.
//
// Replaces dequant + F.linear of the reference (vptq/ops/quant_gemm.py:231-274; kernel
// csrc/kernels/dequant.cuh:9-115), which writes the dense 2 O I bytes of W and reads them back
// through the GEMM for every call.
//
//  * workgroup tile 256 tokens x 128 outputs (16 vector-rows), K step 64 columns, 512 threads =
//    8 waves, one persistent workgroup per CU; consecutive workgroups take the N tiles of the same
//    M block (they share its x rows in L2).
//  * PRODUCER / CONSUMER waves, one of each per SIMD.  Waves 0-3 produce: they dequantise the B
//    tile and issue the A tile's LDS-DMA, and never touch the matrix pipe.  Waves 4-7 consume:
//    each owns a 128 x 64 sub-tile = 4 x 2 v_mfma_f32_32x32x16_{f16,bf16} blocks (128 fp32
//    accumulator registers) and issues nothing but operand reads and 32 MFMAs per K step.  One
//    barrier per K step.  (Earlier versions: every wave did everything in turn - 0.47 PFLOP/s,
//    every gather, global load and LDS round trip exposed; then 8 waves that each filled one tile
//    AND multiplied a 64 x 64 sub-tile - 0.85 PFLOP/s, MFMA pipe busy 40 %: the dequantising
//    wave's ~150 VALU instructions + two LDS round trips + its own 16 MFMAs were the critical
//    path of every K step.)
//  * B tile (W): the 1024 indices of a K step are dequantised by 256 threads (4 consecutive
//    columns of one vector-row each) from the two codebooks in LDS, transposed in registers (4 v_perm_b32 per index) and written IN MFMA-OPERAND
//    ORDER: 16-byte units (8 k values of one output), unit (k chunk, output) at slot
//    chunk * 128 + (output ^ chunk) - the XOR makes both the 8-byte half-unit writes of the
//    dequant threads and the ds_read_b128 operand reads bank-conflict free.
//      fp16: the reference's roundings, w = r16(r16(r16(c + r) * s) + b) - the tile holds the
//            bits vptq_dequant would write; the GEMM accumulates in fp32 like F.linear.
//      bf16: no packed bf16 VALU on gfx950 -> folded form: the tile holds bf16(c + r), the A tile
//            bf16(s * x), and sum_k b_k x[m, k] arrives per token from a small pre-pass
//            (prep_rows_kernel) through the caller's workspace.
//  * A tile (x): LDS-DMA (global_load_lds_dwordx4: no registers, no ds_write): a wave instruction
//    moves 8 tokens x 128 contiguous bytes into 1 KiB of LDS; unit (token, chunk) sits at slot
//    token * 8 + (chunk ^ ((token >> 1) & 7)) - the lanes pick their SOURCE chunk accordingly
//    ("pre-swizzled source") - which makes the ds_read_b128 operand reads conflict free.
//    bf16: the source is bf16(s * x), written once by the pre-pass into the workspace.
//  * both tiles double-buffered: step k's MFMAs read buffer k & 1 while step k + 1 is
//    dequantised / copied into the other one and step k + 2's global loads are in flight; one
//    barrier per K step.  The loop is unrolled by 2 so that buffers and in-flight registers are
//    static (no register copies of loaded values: the compiler counts s_waitcnt vmcnt(n)).
#include <type_traits>

#include "common.h"
#include "kernels.h"
#include "k256.h"

namespace vptq {

constexpr int kFThreads = 512;
constexpr int kFBM = 256, kFBN = 128, kFBK = 64;
constexpr int kFImage = 8192;                           // both codebooks, un-replicated
constexpr int kFTileA = kFBM * kFBK * 2;                // 32 KiB
constexpr int kFTileB = kFBN * kFBK * 2;                // 16 KiB
constexpr int kFNA = 3;                                 // A buffers: DMA two K steps ahead
constexpr int kFLds = kFImage + kFNA * kFTileA + 2 * kFTileB;   // 136 KiB

struct GemmFusedParams {
  const uint32_t* idx;    // [N][row_words]
  const uint32_t* cent;
  const uint32_t* rcent;
  const uint16_t* x;      // [M][G]
  uint16_t* y;            // [M][O]
  const uint16_t* scale;  // [G]
  const uint16_t* wbias;  // [G]
  const uint16_t* bias;   // [O] or NULL
  const float* bx;        // [M] sum_k wbias[k] x[m, k] (bf16 folded form) or NULL
  const uint16_t* xa;     // the A operand: x (fp16) or the pre-scaled copy bf16(s * x) (bf16)
  int M, N, G, O, row_words, tiles_m, tiles_n;
};

typedef _Float16 h8v_t __attribute__((ext_vector_type(8)));
typedef __bf16 b8v_t __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <typename DT>
static __device__ __forceinline__ f32x16 mfma32(u32x4 a, u32x4 b, f32x16 c) {
  if constexpr (std::is_same<DT, F16>::value)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8v_t, a), __builtin_bit_cast(h8v_t, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(b8v_t, a), __builtin_bit_cast(b8v_t, b), c, 0, 0, 0);
}

// bf16 pre-pass, one wave per token row: bx[m] = sum_k b[k] x[m, k] (fp32) and
// xs[m, k] = bf16(s[k] * x[m, k]) (the A operand of the folded form)
template <typename DT>
__global__ __launch_bounds__(256) void prep_rows_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ sc,
                                                        const uint16_t* __restrict__ b, float* __restrict__ bx,
                                                        uint16_t* __restrict__ xs, int M, int G) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  float acc = 0.f;
  for (int c = lane * 8; c < G; c += 512) {
    const u32x4 xv = *(const u32x4*)(x + (size_t)m * G + c), bv = *(const u32x4*)(b + c);
    const u32x4 sv = *(const u32x4*)(sc + c);
    u32x4 o;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      acc = DT::dot2(xv[q], bv[q], acc);
      o[q] = DT::mul2(xv[q], sv[q]);
    }
    *(u32x4*)(xs + (size_t)m * G + c) = o;
  }
  acc = wave_sum(acc);
  if (lane == 0) bx[m] = acc;
}

// Tile order.  Workgroup b runs on XCD b % 8 (observed, not promised: only speed depends on it)
// and every XCD has its own 4 MiB L2.  The tiles, numbered M-major (t = m_block * tiles_n + n_tile),
// are cut into 8 contiguous ranges, one per XCD, which its workgroups walk together: at any time
// the 32 CUs of an XCD multiply N tiles of the SAME M block (or of two), whose x rows (4 MiB per
// 256 tokens x 8192 columns) then stay in that XCD's L2.  Dealing tiles round-robin instead made
// every XCD touch every live M block: x streamed from HBM / Infinity Cache for every N tile
// (8.6 GB per 8192-token layer) and bound the kernel.
struct TileWalk {
  int t, end, step;
  __device__ TileWalk(int n_tiles) {
    const int b = blockIdx.x, g = gridDim.x;
    if (g % 8 == 0) {
      const int xcd = b & 7, w = b >> 3;
      const int cs = (int)((long long)n_tiles * xcd / 8), ce = (int)((long long)n_tiles * (xcd + 1) / 8);
      t = cs + w; end = ce; step = g >> 3;
    } else {
      t = b; end = n_tiles; step = g;
    }
  }
};
static __device__ __forceinline__ void tile_of(int t, int tiles_n, int& tm, int& tn) {
  tm = t / tiles_n;
  tn = t - tm * tiles_n;
}

template <typename DT>
__global__ __launch_bounds__(kFThreads) void gemm_fused_kernel(const GemmFusedParams P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char fsmem[];
  {
    typedef __attribute__((address_space(3))) unsigned char lds_u8_t;
    if ((uint32_t)(uintptr_t)(lds_u8_t*)fsmem != 0u) __builtin_trap();  // absolute LDS addressing
  }
  constexpr bool kF16 = std::is_same<DT, F16>::value;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int G = P.G, M = P.M, O = P.O;
  const uint32_t row_bytes = (uint32_t)P.row_words * 4u;
  const int nk = (G + kFBK - 1) / kFBK;

  // ---- both codebooks in LDS, un-replicated: entry e of the main table at e * 16, of the residual
  // table at 4096 + e * 16.  (The GEMV kernels replicate the tables 8x to make the gathers
  // conflict free; here the 32 gather wave-instructions of a K step are a few per cent of the LDS
  // time, and the 56 KiB buy a third A buffer.)
  {
    const char* const tab = tid < 256 ? (const char*)P.cent : (const char*)P.rcent;
    lds_store16((uint32_t)tid * 16u, *(const u32x4*)as_global(tab + (uint32_t)(tid & 255) * 16u));
  }
  // ---- roles
  const bool producer = __builtin_amdgcn_readfirstlane(wave) < 4;
  // producer, B tile: thread = (vector-row dr of the tile, column quad dq of the K step)
  const int dq = tid & 15, dr = (tid >> 4) & 15;
  const uint32_t dchunk = (uint32_t)dq >> 1, dhalf = (uint32_t)dq & 1u;
  // consumer c = wave - 4: tokens [128 (c >> 1), +128) x outputs [64 (c & 1), +64)
  const int cwv = __builtin_amdgcn_readfirstlane(wave) & 3;
  const int wm = cwv >> 1, wn = cwv & 1;
  const int mi_lane = lane & 31, mkg = lane >> 5;
  // LDS addressing.  A unit (16 bytes = 8 k values of one token / output): every address below is
  // ONE per-lane register + an immediate (block, k step, buffer).
  //   B: unit (chunk c, output o) at slot c * 128 + (o ^ c)            (c < 8: low 3 bits only)
  //   A: unit (token t, chunk c)  at slot t * 8 + (c ^ ((t >> 1) & 7))
  const uint32_t a_buf0 = kFImage, b_buf0 = kFImage + kFNA * kFTileA;
  uint32_t a_rd[4], b_rd[4];   // operand reads of K sub-step ks: + blk * {4096, 512} (+ buffer)
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const uint32_t c = (uint32_t)(ks * 2 + mkg);
    a_rd[ks] = a_buf0 + (uint32_t)(wm * 128 + mi_lane) * 128u + ((c ^ (((uint32_t)mi_lane >> 1) & 7u)) << 4);
    b_rd[ks] = b_buf0 + c * 2048u + (uint32_t)wn * 1024u + (((uint32_t)mi_lane ^ c) << 4);
  }
  // producer, A tile: wave pw moves tokens [64 pw, +64) of a K step with 8 LDS-DMA instructions of
  // 8 tokens x 128 bytes; lane -> (token, slot position p), SOURCE chunk = p ^ swizzle
  // (instruction j, lane l: token = 64 pw + 8 j + (l >> 3), so (token >> 1) & 7 = (l >> 4) + 4 (j & 1))
  const int pw = cwv;
  const int dma_tok = lane >> 3;
  const uint32_t dma_chunk0 = ((uint32_t)lane & 7u) ^ ((uint32_t)lane >> 4);
  // producer, B fill: output dr * 8 + j of chunk dchunk, half dhalf  -> + ((j ^ dchunk) << 4) + buf * tile
  const uint32_t b_wr = b_buf0 + dchunk * 2048u + (uint32_t)dr * 128u + dhalf * 8u;

  __syncthreads();  // codebooks in place

  auto ring_next = [](uint32_t off) -> uint32_t { return off + kFTileA >= kFNA * kFTileA ? 0u : off + kFTileA; };
  // The role branch sits OUTSIDE the tile loop: inside it, the loop invariants of BOTH roles were
  // hoisted above the branch and stayed live through the other role's K loop (367 spills).
  const int n_tiles = P.tiles_m * P.tiles_n;
  if (producer) {
    for (TileWalk tw(n_tiles); tw.t < tw.end; tw.t += tw.step) {
      int tm, tn;
      tile_of(tw.t, P.tiles_n, tm, tn);
      const int n0 = tn * (kFBN / 8);   // first vector-row
      // ================= producer =================
      struct StageB { u32x2 iw, sv, bv; };
      const int drow = n0 + dr < P.N ? n0 + dr : P.N - 1;
      auto load_b = [&](int ks, StageB& st) {
        const int k0 = (ks < nk ? ks : nk - 1) * kFBK;
        const int dcol = k0 + dq * 4 < G ? k0 + dq * 4 : G - 4;
        st.iw = __builtin_nontemporal_load((const u32x2*)as_global((const char*)P.idx + (size_t)drow * row_bytes + (size_t)dcol * 2u));
        st.sv = *(const u32x2*)as_global(P.scale + dcol);
        st.bv = *(const u32x2*)as_global(P.wbias + dcol);
      };
      // dequantise 4 indices -> 4 columns x 8 outputs (all 8 gathers in flight), transpose, write
      auto fill_b = [&](int ks, int buf, const StageB& st) {
        const int k0 = ks * kFBK;
        const bool dvalid = k0 + dq * 4 < G;
#ifndef VPTQ_FUSED_DBG
#define VPTQ_FUSED_DBG 0   // timing experiments only: 1 = no arithmetic, 2 = no gathers either, 3 = no MFMA
#endif
        u32x4 cA[4], cB[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const uint32_t w = st.iw[u >> 1];
          if (VPTQ_FUSED_DBG == 2) { cA[u] = u32x4{w, w, w, w}; cB[u] = cA[u]; continue; }
          cA[u] = lds_load16(__builtin_amdgcn_ubfe(w, (uint32_t)(16 * (u & 1)), 8u) << 4);
          cB[u] = lds_load16(4096u + (__builtin_amdgcn_ubfe(w, (uint32_t)(16 * (u & 1) + 8), 8u) << 4));
        }
        uint32_t w2[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int p = 0; p < 4; ++p) {
            uint32_t v = VPTQ_FUSED_DBG ? (cA[u][p] ^ cB[u][p]) : DT::add2(cA[u][p], cB[u][p]);
            if constexpr (kF16 && !VPTQ_FUSED_DBG) {
              v = DT::mul2_bcast(v, st.sv[u >> 1], u & 1);
              v = DT::add2_bcast(v, st.bv[u >> 1], u & 1);
            }
            w2[u][p] = dvalid ? v : 0u;
          }
        typedef __attribute__((address_space(3))) u32x2 lds_w_u32x2_t;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          // outputs 2p (low halves) and 2p + 1 (high halves) of the thread's vector-row
          const u32x2 lo = {__builtin_amdgcn_perm(w2[1][p], w2[0][p], 0x05040100u),
                            __builtin_amdgcn_perm(w2[3][p], w2[2][p], 0x05040100u)};
          const u32x2 hi2 = {__builtin_amdgcn_perm(w2[1][p], w2[0][p], 0x07060302u),
                             __builtin_amdgcn_perm(w2[3][p], w2[2][p], 0x07060302u)};
          *(lds_w_u32x2_t*)(uintptr_t)(b_wr + (((uint32_t)(2 * p) ^ dchunk) << 4) + (uint32_t)buf * kFTileB) = lo;
          *(lds_w_u32x2_t*)(uintptr_t)(b_wr + (((uint32_t)(2 * p + 1) ^ dchunk) << 4) + (uint32_t)buf * kFTileB) = hi2;
        }
      };
      // Step k (consumers multiply step k): fill B buffer (k + 1) & 1, request the B inputs of step
      // k + 3.  Loop unrolled by 2: static B buffers and stage registers, counted vmcnt waits.
      StageB st[2];
      load_b(0, st[0]);
      load_b(1, st[1]);
      __syncthreads();           // the previous tile's MFMA reads are done
      fill_b(0, 0, st[0]);
      load_b(2, st[0]);
      __syncthreads();
      for (int k = 0; k < nk; k += 2) {
        if (k + 1 < nk) fill_b(k + 1, 1, st[1]);
        load_b(k + 3, st[1]);
        __syncthreads();
        if (k + 1 >= nk) break;
        if (k + 2 < nk) fill_b(k + 2, 0, st[0]);
        load_b(k + 4, st[0]);
        __syncthreads();
      }
    }
  } else {
    for (TileWalk tw(n_tiles); tw.t < tw.end; tw.t += tw.step) {
      int tm, tn;
      tile_of(tw.t, P.tiles_n, tm, tn);
      const int m0 = tm * kFBM;
      // ================= consumer =================
      f32x16 acc[4][2];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
      // one K step: 4 sub-steps of K = 16, each 4 A + 2 B operand reads and 8 MFMAs; the operands
      // of sub-step ks + 1 are requested before the MFMAs of sub-step ks
      auto mma = [&](uint32_t b_off, uint32_t a_off) {
        u32x4 av[2][4], bv4[2][2];
        auto rd = [&](int ks, int sl) {
          const uint32_t ar = a_rd[ks] + a_off, br = b_rd[ks] + b_off;
#pragma unroll
          for (int a = 0; a < 4; ++a) av[sl][a] = lds_load16(ar + (uint32_t)a * 4096u);
#pragma unroll
          for (int b = 0; b < 2; ++b) bv4[sl][b] = lds_load16(br + (uint32_t)b * 512u);
        };
        rd(0, 0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          // (fenced: left alone the scheduler requests all four sub-steps' operands up front -
          // 96 registers beside the 128 accumulators - and spills the accumulators)
          __builtin_amdgcn_sched_barrier(0);
          if (ks + 1 < 4) rd(ks + 1, (ks + 1) & 1);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
              if (VPTQ_FUSED_DBG == 3) { acc[a][b][0] += __uint_as_float(av[ks & 1][a][0] ^ bv4[ks & 1][b][0]); continue; }
              acc[a][b] = mfma32<DT>(av[ks & 1][a], bv4[ks & 1][b], acc[a][b]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
      };
      // A tile of K step ks -> ring slot a_off by LDS-DMA, tokens [64 pw, +64): 8 instructions of
      // 8 tokens x 128 bytes.  Issued as inline assembly with hand-counted waits: an LDS-DMA the
      // compiler can see makes it wait for ALL memory operations (vmcnt(0)) before the next LDS
      // read - every K step then paid the full latency of the A tile two steps ahead.  This wave
      // has no other memory loads inside the K loop, so the count is exact: 8 per step.
      // (Columns past G land as whatever the clamped source holds - finite x values - and meet
      // zeros in the B tile.)
      uint32_t dma_voff[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int token = pw * 64 + 8 * j + dma_tok;
        const int mrow = m0 + token < M ? m0 + token : M - 1;
        dma_voff[j] = (uint32_t)(mrow - m0) * (uint32_t)G * 2u;
      }
      const char* const xa_tile = (const char*)P.xa + (size_t)m0 * G * 2u;
      auto dma_a = [&](int ks, uint32_t a_off) {
        const int k0 = (ks < nk ? ks : nk - 1) * kFBK;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int chunk = (int)(dma_chunk0 ^ (uint32_t)(4 * (j & 1)));
          const int col = k0 + chunk * 8 < G ? k0 + chunk * 8 : G - 8;
          const uint32_t voff = dma_voff[j] + (uint32_t)col * 2u;
          const uint32_t dst = a_buf0 + a_off + (uint32_t)(pw * 64 + 8 * j) * 128u;
          asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2"
                       :: "s"(dst), "v"(voff), "s"(xa_tile) : "memory", "m0");
        }
      };
      __syncthreads();           // the previous tile's MFMA reads are done
      dma_a(0, 0);
      dma_a(1, kFTileA);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // step 0 has landed; step 1 may still fly
      __syncthreads();
      uint32_t a_off = 0, b_off = 0, a_off2 = 2 * kFTileA;
      for (int k = 0; k < nk; ++k) {   // ONE call site: the 128 accumulator registers stay put
        dma_a(k + 2, a_off2);          // into the slot step k - 1 used (past the end: a harmless re-read)
        a_off2 = ring_next(a_off2);
        mma(b_off, a_off);
        a_off = ring_next(a_off);
        b_off ^= (uint32_t)kFTileB;
        // step k + 1 must have landed before the barrier publishes it; step k + 2 keeps flying
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __syncthreads();
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      // ---- epilogue: D block (a, b): row (token) = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5),
      // column (output) = lane & 31
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const int o = tn * kFBN + wn * 64 + b * 32 + (lane & 31);
          const float ob = (P.bias && o < O) ? DT::to_float(as_global(P.bias)[o]) : 0.f;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * 128 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (m < M && o < O) {
              float v = acc[a][b][r] + ob;
              if constexpr (!kF16) v += P.bx[m];
              as_global(P.y)[(size_t)m * O + o] = DT::from_float(v);
            }
          }
        }
    }
  }
}

// ---- host side -------------------------------------------------------------------
bool gemm_fused_eligible(const VptqLayerDesc& d) {
  if (!gemv_k256_eligible(d, 1)) return false;          // canonical format, norm on, aligned
  if (d.perm) return false;                             // (absorb_perm first)
  return d.group_size >= 64 && (d.group_size & 7) == 0;
}

// bf16: [bx: tokens floats, padded to 256 bytes][xs: tokens x I bf16]
static size_t bx_bytes(int tokens) { return ((size_t)tokens * sizeof(float) + 255) / 256 * 256; }
size_t gemm_fused_workspace_bytes(const VptqLayerDesc& d, int tokens) {
  return d.dtype == VPTQ_DTYPE_F16 ? 0 : bx_bytes(tokens) + (size_t)tokens * d.in_features * 2;
}

hipError_t launch_gemm_fused(const VptqLayerDesc& d, const void* x, void* y, int tokens, void* workspace,
                             size_t workspace_bytes, hipStream_t st) {
  GemmFusedParams P = {};
  P.idx = (const uint32_t*)d.indices;
  P.cent = (const uint32_t*)d.centroids;
  P.rcent = (const uint32_t*)d.res_centroids;
  P.x = (const uint16_t*)x;
  P.y = (uint16_t*)y;
  P.scale = (const uint16_t*)d.weight_scale;
  P.wbias = (const uint16_t*)d.weight_bias;
  P.bias = (const uint16_t*)d.bias;
  P.M = tokens; P.N = d.num_indices; P.G = d.group_size; P.O = d.out_features; P.row_words = d.row_words;
  P.tiles_m = (tokens + kFBM - 1) / kFBM;
  P.tiles_n = (d.out_features + kFBN - 1) / kFBN;
  const bool f16 = d.dtype == VPTQ_DTYPE_F16;
  P.xa = P.x;
  if (!f16) {
    if (!workspace || workspace_bytes < gemm_fused_workspace_bytes(d, tokens) || (((uintptr_t)workspace) & 15))
      return hipErrorInvalidValue;
    P.bx = (const float*)workspace;
    uint16_t* xs = (uint16_t*)((char*)workspace + bx_bytes(tokens));
    P.xa = xs;
    hipLaunchKernelGGL((prep_rows_kernel<BF16>), dim3((tokens + 3) / 4), dim3(256), 0, st, P.x, P.scale, P.wbias,
                       (float*)workspace, xs, tokens, d.group_size);
    if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
  }
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  static std::atomic<int> cus[64];
  if (!cus[dev]) {
    hipDeviceProp_t p;
    cus[dev] = hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
  }
  static std::atomic<bool> attr_set[64];
  if (!attr_set[dev]) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_fused_kernel<F16>, hipFuncAttributeMaxDynamicSharedMemorySize, kFLds);
    if (e == hipSuccess)
      e = hipFuncSetAttribute((const void*)gemm_fused_kernel<BF16>, hipFuncAttributeMaxDynamicSharedMemorySize, kFLds);
    if (e != hipSuccess) return e;
    attr_set[dev] = true;
  }
  const int n_tiles = P.tiles_m * P.tiles_n;
  // one workgroup per CU; kept a multiple of 8 (the XCD count) for the tile walk
  const int ncu = cus[dev].load();
  int grid = n_tiles < ncu ? (n_tiles + 7) / 8 * 8 : ncu;
  if (grid > cus[dev]) grid = cus[dev];
  if (f16) hipLaunchKernelGGL((gemm_fused_kernel<F16>), dim3(grid), dim3(kFThreads), kFLds, st, P);
  else hipLaunchKernelGGL((gemm_fused_kernel<BF16>), dim3(grid), dim3(kFThreads), kFLds, st, P);
  return hipGetLastError();
}

}  // namespace vptq
