// Batched decode for the canonical "2-bit" VPTQ format (v = 8, 256 + 256 centroids, norm on):
// up to 16 tokens in ONE launch, tokens = the M dimension of a real matrix-core contraction.
//
// Reference: vptq/ops/quant_gemm.py:213-274 - its fused GEMV takes < 3 tokens, everything else
// is dequant to a dense W + F.linear; its v2 kernel takes < 16 (csrc/quant_gemv_v2.cu:58).
// Round 1 served 5-16 tokens as launches of <= 4 tokens of gemv_k256m_kernel (48 us per
// 8192^2 layer at 16 tokens); the 4x4x4 MFMA of that kernel multiplies 4 tokens at most.
//
// Structure (one 1024-thread workgroup per CU, persistent over groups of 4 vector-rows):
//  * the 64 KiB conflict-free codebook image of gemv_k256m.hip (8 replicas per table, the
//    two gathers of an index split across the lanes).
//  * per step a TILE of 4 vector-rows x 1024 columns (4096 indices, 4 per thread) is
//    dequantised with the reference CPU path's roundings, w = r16(r16(r16(c + r) * s) + b)
//    in packed f16 - bit-identical to vptq_dequant - and written to LDS IN MFMA-OPERAND ORDER:
//    a thread owns 4 consecutive columns of one vector-row, so after an in-register 4 x 8
//    transposition (4 v_perm_b32 per index) it holds, for each of its 8 outputs, the 4
//    consecutive-k values one lane of v_mfma_f32_16x16x16_f16 supplies as B operand: one
//    64-byte run of the tile per thread, written with 4 ds_write_b128 whose 16-byte column is
//    XOR-swizzled with the k position so that the 8 lanes of a write group hit 8 different
//    bank quads; the B reads (ds_read_b64, [k group][output] contiguous) are conflict free.
//  * MFMA phase: the 64 K-steps of a tile are dealt to the 16 waves (4 each, both 16-output
//    blocks): D[token][output] += X[token][k] * W[output][k], fp32.  The A operands (raw x, 8
//    bytes per lane and K-step) are loaded from L2 once per column block and reused for every
//    row group of the workgroup (columns outer, row groups inner).
//  * at the end the 16 waves' partial D are summed through LDS and stored (+ output bias).
// Per index: 2 address perms + 12 packed-f16 ops + 4 transposition perms on the VALU, 2
// gathers + 16 bytes of LDS writes + 16 bytes of LDS operand reads; the MFMA work is 1 / 16 of
// the instruction stream.  bf16 (round 6): the same tile, its roundings as blocks of v_dot2_f32_bf16 (common.h: BF16::add4 /
// scale_bias4 - 44 instructions per index instead of 12) and v_mfma_f32_16x16x16_bf16; before, bf16 kept the <= 4-token launches.
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "kernels.h"
#include "k256.h"

namespace vptq {

constexpr int kGThreads = 1024;
constexpr int kGWaves = 16;
constexpr int kGRows = 4;            // vector-rows per row group (32 outputs = 2 MFMA N blocks)
constexpr int kGTileCols = 1024;     // columns per tile
constexpr int kGImage = 65536;
constexpr int kGTile = kGRows * 8 * kGTileCols * 2;  // 64 KiB
constexpr int kGLds = kGImage + kGTile;

struct GemmK256Params {
  const uint32_t* idx;    // [N][row_words]
  const uint32_t* cent;
  const uint32_t* rcent;
  const uint16_t* x;      // [tokens][G]
  void* y;                // [tokens][O]
  const uint16_t* scale;  // [G] column order
  const uint16_t* wbias;  // [G] column order
  const uint16_t* bias;   // [O] or NULL
  const uint16_t* perm;   // [G] or NULL
  int N, G, O, row_words, tokens, out_f32, n_groups;
};

typedef _Float16 h4v_t __attribute__((ext_vector_type(4)));

// Tile geometry.  A tile is 4 vector-rows (2 blocks nb of 16 outputs) x 1024 columns = 256
// column quads.  MFMA wave wq consumes quads [16 wq, 16 wq + 16): local quad Q = kg * 4 + i is
// K-step i, lane group kg - so that a lane's four K-steps are 16 CONSECUTIVE columns (its x
// operands are two 16-byte loads of one 128-byte line per token; with K-steps of 16 adjacent
// columns a lane read four 8-byte pieces of four different lines, 4x the L2 traffic - that
// was 80 % of the first version's time).  Operand row R = ((nb * 16 + wq) * 4 + i) * 4 + kg:
// 128 bytes = 16 outputs x 4 k values; kg fastest, so the 32 lanes of half a ds_read_b64
// read 256 contiguous bytes.  Inside a row the 16-byte column is XOR-ed with i: the 8 lanes
// of a ds_write_b128 group (4 consecutive quads x the two rows of nb) then cover all 32 banks.
static __device__ __forceinline__ uint32_t tile_row(int nb, int wq, int i, int kg) {
  return (uint32_t)((((nb * 16 + wq) * 4 + i) * 4) + kg);
}

template <typename DT, bool PERM, int NRG>
static __device__ __forceinline__ void gemm_k256_pass(const GemmK256Params& P, unsigned char* gsmem,
                                                      const int rg0) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int G = P.G, N = P.N, O = P.O, tokens = P.tokens;
  const uint32_t row_bytes = (uint32_t)P.row_words * 4u;
  const int n_tiles = (G + kGTileCols - 1) / kGTileCols;
  const int grid = (int)gridDim.x;

  const uint32_t hi = (lane >> 3) & 1u;
  const uint32_t baseA = ((hi << 3) | (lane & 7u)) << 4;
  const uint32_t baseB = (((hi ^ 1u) << 3) | (lane & 7u)) << 4;
  const uint32_t selGA[2] = {0x0c0c0400u | (hi << 8), 0x0c0c0600u | (hi << 8)};
  const uint32_t selGB[2] = {0x0c0c0400u | ((hi ^ 1u) << 8), 0x0c0c0600u | ((hi ^ 1u) << 8)};

  // ---- dequant role: thread = 4 consecutive columns (one quad) of one vector-row
  const int dnb = wave >> 3;                                        // 16-output block
  const int drp = (lane >> 2) & 1;                                  // row of the block
  const int drow = dnb * 2 + drp;                                   // vector-row inside the group
  const int dq = (wave & 7) * 32 + (lane >> 3) * 4 + (lane & 3);    // quad inside the tile
  const int di = dq & 3, dkg = (dq >> 2) & 3, dwq = dq >> 4;
  const uint32_t tile_wr = kGImage + tile_row(dnb, dwq, di, dkg) * 128u + (uint32_t)drp * 64u;
  // ---- MFMA role: wave wq = wave, lane (output j / token, k group)
  const int mj = lane & 15, mkg = lane >> 4;

  f32x4 acc[NRG][2];
#pragma unroll
  for (int g = 0; g < NRG; ++g) { acc[g][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[g][1] = acc[g][0]; }

  // A pass is a flat sequence of steps (tile, row group), row groups fastest; the step loop is
  // unrolled by kQ = 4 (NRG divides it), so every register-queue slot, the row group and "is
  // this the first / last row group of its tile" are compile-time facts of each unrolled
  // body, and every body issues a fixed, unconditional set of loads: the compiler counts
  // them (s_waitcnt vmcnt(n), never 0) and nothing waits for a load it has just issued.
  constexpr int kQ = 4;
  const int n_steps = n_tiles * NRG;
  auto load_idx = [&](int step) -> u32x2 {
    const int sc = step < n_steps ? step : n_steps - 1;   // past the end: the last step again (unused)
    const int tile = sc / NRG, g = sc % NRG;
    const int row = (rg0 + g * grid) * kGRows + drow;
    const int col = tile * kGTileCols + dq * 4;
    const uint32_t roff = (uint32_t)(row < N ? row : N - 1) * row_bytes;
    const uint32_t coff = (uint32_t)(col < G ? col : G - 4) * 2u;
    return __builtin_nontemporal_load((const u32x2*)as_global((const char*)P.idx + (size_t)roff + coff));
  };
  const uint16_t* const xrow = as_global(P.x + (size_t)(mj < tokens ? mj : tokens - 1) * G);
  auto load_x = [&](int tile, u32x4& lo, u32x4& hi4) {
    const int col = tile * kGTileCols + wave * 64 + mkg * 16;
    const int c0 = col < G ? col : G - 8, c1 = col + 8 < G ? col + 8 : G - 8;
    if constexpr (PERM) {
      const u32x4 p0 = *(const u32x4*)as_global(P.perm + c0), p1 = *(const u32x4*)as_global(P.perm + c1);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        lo[q] = (uint32_t)xrow[p0[q] & 0xffffu] | ((uint32_t)xrow[p0[q] >> 16] << 16);
        hi4[q] = (uint32_t)xrow[p1[q] & 0xffffu] | ((uint32_t)xrow[p1[q] >> 16] << 16);
      }
    } else {
      lo = *(const u32x4*)(xrow + c0);
      hi4 = *(const u32x4*)(xrow + c1);
    }
  };
  auto load_sb = [&](int tile, u32x2& sv, u32x2& bv) {
    const int dcol = tile * kGTileCols + dq * 4;
    const int dcc = dcol < G ? dcol : G - 4;
    sv = *(const u32x2*)as_global(P.scale + dcc);
    bv = *(const u32x2*)as_global(P.wbias + dcc);
  };
  // (scale / bias first: the loop waits for them with the index words still in flight)
  u32x2 sv, bv, sv_next, bv_next;
  load_sb(0, sv_next, bv_next);
  __builtin_amdgcn_sched_barrier(0);
  u32x2 iq[kQ];
#pragma unroll
  for (int k = 0; k < kQ; ++k) iq[k] = load_idx(k);
  u32x4 xa_lo = {0, 0, 0, 0}, xa_hi = {0, 0, 0, 0};

  auto do_step = [&](auto kc, int step) {
    constexpr int K = decltype(kc)::value;
    constexpr int g = K % NRG;
    const int tile = step / NRG;
    const u32x2 iw = iq[K];
    if constexpr (g == 0) {
      // new column block: scale / bias requested during the previous block; x of this block
      // (used in the MFMA phase, ~1000 cycles from here)
      sv = sv_next; bv = bv_next;
      __builtin_amdgcn_sched_barrier(0);
      load_x(tile, xa_lo, xa_hi);
    }
    if constexpr (g == NRG - 1) load_sb(tile + 1 < n_tiles ? tile + 1 : tile, sv_next, bv_next);
    __builtin_amdgcn_sched_barrier(0);
    const bool dvalid = tile * kGTileCols + dq * 4 < G;
    // ---- dequantise 4 indices: gathers, c + r, * s, + b (each rounded to f16)
    uint32_t w2[4][4];
#pragma unroll
    for (int u0 = 0; u0 < 4; u0 += 2) {   // two columns (one index word) at a time: registers
      u32x4 cA[2], cB[2];
      const uint32_t w = iw[u0 >> 1];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        cA[h] = lds_load16(__builtin_amdgcn_perm(w, baseA, selGA[h]));
        cB[h] = lds_load16(__builtin_amdgcn_perm(w, baseB, selGB[h]));
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        // (four packed pairs per call: for bf16 each stage is one scheduled block of dot instructions, common.h)
        uint32_t v[4] = {cA[h][0], cA[h][1], cA[h][2], cA[h][3]};
        const uint32_t r[4] = {cB[h][0], cB[h][1], cB[h][2], cB[h][3]};
        DT::add4(v, r);
        DT::scale_bias4(v, sv[u0 >> 1], h, bv[u0 >> 1], h);
#pragma unroll
        for (int p = 0; p < 4; ++p) w2[u0 + h][p] = dvalid ? v[p] : 0u;
      }
    }
    // this slot's index words are consumed: request the step kQ ahead into it
    __builtin_amdgcn_sched_barrier(0);
    iq[K] = load_idx(step + kQ);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();  // the previous step's MFMA reads of the tile are done
    // ---- transpose 4 columns x 8 outputs -> 8 operands of 4 k values; 64 bytes per thread
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const u32x4 v = {__builtin_amdgcn_perm(w2[1][p], w2[0][p], 0x05040100u),
                       __builtin_amdgcn_perm(w2[3][p], w2[2][p], 0x05040100u),
                       __builtin_amdgcn_perm(w2[1][p], w2[0][p], 0x07060302u),
                       __builtin_amdgcn_perm(w2[3][p], w2[2][p], 0x07060302u)};
      lds_store16(tile_wr + (((uint32_t)p ^ (uint32_t)di) << 4), v);
    }
    __syncthreads();  // tile complete
    // ---- MFMA: this wave's 64 columns = 4 K-steps x 2 output blocks, into row group g
    const int colw = tile * kGTileCols + wave * 64 + mkg * 16;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool live = mj < tokens && colw + 4 * i < G;
      const u32x4& xs4 = i < 2 ? xa_lo : xa_hi;
      const u32x2 av = {live ? xs4[(i & 1) * 2] : 0u, live ? xs4[(i & 1) * 2 + 1] : 0u};
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        const uint32_t pair = ((uint32_t)mj >> 1) ^ (uint32_t)i;
        typedef __attribute__((address_space(3))) const u32x2 lds_u32x2_t;
        const u32x2 b = *(lds_u32x2_t*)(uintptr_t)(kGImage + tile_row(nb, wave, i, mkg) * 128u + pair * 16u +
                                                   ((uint32_t)mj & 1u) * 8u);
        if constexpr (std::is_same<DT, F16>::value)
          acc[g][nb] = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(h4v_t, av),
                                                             __builtin_bit_cast(h4v_t, b), acc[g][nb], 0, 0, 0);
        else
          acc[g][nb] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s4_t, av),
                                                                 __builtin_bit_cast(s4_t, b), acc[g][nb], 0, 0, 0);
      }
    }
  };
  for (int step = 0; step < n_steps; step += kQ) {
    do_step(std::integral_constant<int, 0>{}, step);
    if (step + 1 < n_steps) do_step(std::integral_constant<int, 1>{}, step + 1);
    if (step + 2 < n_steps) do_step(std::integral_constant<int, 2>{}, step + 2);
    if (step + 3 < n_steps) do_step(std::integral_constant<int, 3>{}, step + 3);
  }
  // ---- sum the 16 waves' partial D and store, one row group at a time (tile area as scratch)
  float* const scr = (float*)(gsmem + kGImage);  // [nb][reg][wave][lane]
#pragma unroll
  for (int g = 0; g < NRG; ++g) {
    __syncthreads();
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r) scr[((nb * 4 + r) * kGWaves + wave) * 64 + lane] = acc[g][nb][r];
    __syncthreads();
    if (tid < 512) {
      // D element (token = (l >> 4) * 4 + r, output = l & 15) of block nb
      const int nb = tid >> 8, r = (tid >> 6) & 3, l = tid & 63;
      float sum = 0.f;
#pragma unroll
      for (int w = 0; w < kGWaves; ++w) sum += scr[((nb * 4 + r) * kGWaves + w) * 64 + l];
      const int token = (l >> 4) * 4 + r;
      const int o = (rg0 + g * grid) * (kGRows * 8) + nb * 16 + (l & 15);
      if (token < tokens && o < O) {
        if (P.bias) sum += DT::to_float(as_global(P.bias)[o]);
        if (P.out_f32) ((float*)as_global((uint16_t*)P.y))[(size_t)token * O + o] = sum;
        else as_global((uint16_t*)P.y)[(size_t)token * O + o] = DT::from_float(sum);
      }
    }
  }
  __syncthreads();
}

template <typename DT, bool PERM>
__global__ __launch_bounds__(kGThreads) void gemm_k256_kernel(const GemmK256Params P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char gsmem[];
  {
    typedef __attribute__((address_space(3))) unsigned char lds_u8_t;
    if ((uint32_t)(uintptr_t)(lds_u8_t*)gsmem != 0u) __builtin_trap();  // absolute LDS addressing
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // ---- codebook image (see gemv_k256m.hip): row e = 8 main replicas | 8 residual replicas
  {
    const char* const tab = __builtin_amdgcn_readfirstlane(wave) < 8 ? (const char*)P.cent : (const char*)P.rcent;
    const u32x4 centry = *(const u32x4*)as_global(tab + (uint32_t)((tid >> 1) & 255) * 16u);
    const uint32_t rowp = ((uint32_t)((tid >> 1) & 255) << 8) | ((uint32_t)(tid >> 9) << 7) |
                          ((uint32_t)(tid & 1) << 6);
#pragma unroll
    for (int q = 0; q < 4; ++q) lds_store16(rowp + (((q + (lane >> 1)) & 3) << 4), centry);
  }
  __syncthreads();  // image complete
  // this workgroup's row groups bid, bid + grid, ...: passes of 4, 2, 1 (the x operands of a
  // column block are loaded once per pass and reused for every row group of the pass)
  const int grid = (int)gridDim.x;
  int left = ((int)P.n_groups - (int)blockIdx.x + grid - 1) / grid;
  int rg = blockIdx.x;
  for (; left >= 4; left -= 4, rg += 4 * grid) gemm_k256_pass<DT, PERM, 4>(P, gsmem, rg);
  if (left >= 2) { gemm_k256_pass<DT, PERM, 2>(P, gsmem, rg); left -= 2; rg += 2 * grid; }
  if (left >= 1) gemm_k256_pass<DT, PERM, 1>(P, gsmem, rg);
}

// ---- host side -------------------------------------------------------------------
bool gemm_k256_eligible(const VptqLayerDesc& d, int tokens, int flags) {
  (void)flags;
  if (!gemv_k256_eligible(d, 1)) return false;      // canonical format, norm on, aligned
  if (d.group_size < 8 || (d.group_size & 7)) return false;
  if (d.perm && !(d.scale_permuted && d.bias_permuted)) return false;
  return tokens >= 1 && tokens <= 16;
}

hipError_t launch_gemm_k256(const VptqLayerDesc& d, const void* x, void* y, int tokens, bool out_f32,
                            hipStream_t st) {
  GemmK256Params P = {};
  P.idx = (const uint32_t*)d.indices;
  P.cent = (const uint32_t*)d.centroids;
  P.rcent = (const uint32_t*)d.res_centroids;
  P.x = (const uint16_t*)x;
  P.y = y;
  P.scale = (const uint16_t*)(d.perm ? d.scale_permuted : d.weight_scale);
  P.wbias = (const uint16_t*)(d.perm ? d.bias_permuted : d.weight_bias);
  P.bias = (const uint16_t*)d.bias;
  P.perm = d.perm;
  P.N = d.num_indices; P.G = d.group_size; P.O = d.out_features; P.row_words = d.row_words;
  P.tokens = tokens; P.out_f32 = out_f32 ? 1 : 0;
  P.n_groups = (d.num_indices + kGRows - 1) / kGRows;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  static std::atomic<int> cus[64];
  if (!cus[dev]) {
    hipDeviceProp_t p;
    cus[dev] = hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
  }
  const int ncu = cus[dev].load();
  const int grid = P.n_groups < ncu ? P.n_groups : ncu;
  auto go = [&](auto kern, std::atomic<bool>& done) -> hipError_t {
    if (!done) {
      if (hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, kGLds); e != hipSuccess) return e;
      done = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kGThreads), kGLds, st, P);
    return hipGetLastError();
  };
  static std::atomic<bool> attr_set[4][64];
  if (d.dtype == VPTQ_DTYPE_F16)
    return d.perm ? go(gemm_k256_kernel<F16, true>, attr_set[0][dev]) : go(gemm_k256_kernel<F16, false>, attr_set[1][dev]);
  return d.perm ? go(gemm_k256_kernel<BF16, true>, attr_set[2][dev]) : go(gemm_k256_kernel<BF16, false>, attr_set[3][dev]);
}

}  // namespace vptq
