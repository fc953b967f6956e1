// Fused dequant + GEMV for the canonical "2-bit" VPTQ format on MI355X (gfx950):
//   v = 8, k = 256 main + 256 residual centroids (T = 16 bits / index pair),
//   one codebook, no outlier columns, weight_scale/weight_bias present.
//
// Replaces WqA16WithOutliers_PackIndice + tmp.sum(-1)
// (reference csrc/kernels/quant_gemv.cuh:11-186, csrc/quant_gemv.cu:203-235).
//
// Design (numbers from tools/ubench.hip on MI355X):
//  * the kernel is VALU-issue bound, not HBM bound: rebuilding 8 weights with the
//    reference's three roundings costs 22 VALU instructions per 2-byte index pair
//    and a SIMD issues one packed-f16 / fma instruction per ~2.3 ns only with
//    >= 4 waves resident (5.1 ns with one).  Everything below serves occupancy
//    and instruction count:
//  * 512-thread workgroups, 2 per CU (4 waves / SIMD, <= 128 VGPRs), each owning
//    ROWS complete vector-rows (8*ROWS outputs) over ALL input columns -> no
//    split-K buffer, no second kernel;
//  * every lane streams 16-byte pieces of the packed index rows (8 column indices
//    per load, 1 KiB per wave instruction, fully coalesced), all loads of an
//    iteration issued before the first use;
//  * both codebooks live in ONE 64 KiB LDS image of 256 bank rows: row e holds
//    8 replicas of main entry e (slots 0-7) and 8 replicas of residual entry e
//    (slots 8-15).  The two gathers of an index are split across the lanes: lanes with
//    bit 3 clear fetch the main entry first (slot l & 7) and the residual entry second
//    (slot 8 + (l & 7)), the others the other way round, so every 16-lane group of a
//    ds_read_b128 touches 16 different slots - conflict free for ANY index pattern
//    (all lanes on one table: 2-way conflicts, 8.8 instead of 5 LDS cycles; a plain
//    4 KiB table: ~3x serialisation).  Which register holds which entry does not
//    matter, the two are added (exact form) or accumulated with the same x (folded);
//  * the LDS address of a gather is ONE v_perm_b32 (index byte -> bits 8..15,
//    lane slot / table -> bits 4..7);
//  * weights are rebuilt with the reference CPU path's roundings
//    (r16(r16(r16(c+r)*s)+b), packed f16 VALU with op_sel broadcasts) and x*w is
//    accumulated in fp32 by v_fma_mix_f32;
//  * the 8*ROWS*TOK partial sums of a lane are reduced over the wave with the
//    gfx950 lane-swap instructions (~2 ops per value instead of 12);
//  * two entry points share the body: gemv_k256_kernel (layer = blockIdx.y, grouped
//    launches, 2-4 tokens) and gemv_k256_kernel_1 (one layer, one token: the arguments the
//    first loads need are preloaded into SGPRs at wave launch).
#include "common.h"
#include "kernels.h"
#include "k256.h"

namespace vptq {

constexpr int kThreads = 512;
constexpr int kWaves = kThreads / 64;
constexpr int kSweepCols = kThreads * 8;  // columns covered by one sweep of the WG
// both gathers of one element: index byte h*2 -> main entry, byte h*2+1 -> residual
// The two gathers are split ACROSS the lanes: in the first, lanes with bit 3 clear fetch the
// main entry (slot lane & 7) and the others the residual entry (slot 8 + (lane & 7)); the second
// is the complement.  Every 16-lane group of a ds_read_b128 then touches 16 different slots
// (conflict free: 5 instead of 8.8 LDS cycles per gather, tools/ubench_lds.hip), and since the
// two entries are only ever added it does not matter which register holds which.
static __device__ __forceinline__ void gather(uint32_t w, int h, uint32_t baseC, uint32_t baseR,
                                              uint32_t selC, uint32_t selR, u32x4& cv, u32x4& rv) {
  // D = {0, 0, w.byte, base.b0}: (index << 8) | slot; element h of the word = bytes 2h, 2h + 1
  const uint32_t aC = __builtin_amdgcn_perm(w, baseC, h ? selC + 0x200u : selC);
  const uint32_t aR = __builtin_amdgcn_perm(w, baseR, h ? selR + 0x200u : selR);
  cv = lds_load16(aC);
  rv = lds_load16(aR);
}

// 16-byte load of 8 consecutive 16-bit values, or (PERM) a gather of 8 values through
// the permutation: column c of the quantised matrix multiplies input feature perm[c].
// 16 bytes at base + byte_off: 32-bit zero-extended offset against a wave-uniform base
// selects the scalar-base addressing form (no 64-bit vector add per load).
static __device__ __forceinline__ u32x4 ld16(const void* base, uint32_t byte_off) {
  return *(const u32x4*)as_global((const char*)base + byte_off);
}

template <bool PERM>
static __device__ __forceinline__ u32x4 load8(const uint16_t* __restrict__ p, int col0,
                                              const u32x4& pv) {
  if (!PERM) return ld16(p, (uint32_t)col0 * 2u);
  const uint16_t* const g = as_global(p);
  u32x4 r;
#pragma unroll
  for (int q = 0; q < 4; ++q)
    r[q] = (uint32_t)g[pv[q] & 0xffffu] | ((uint32_t)g[pv[q] >> 16] << 16);
  return r;
}

template <typename DT, int ROWS, int TOK, int SW, bool PERM, bool FAST>
static __device__ __forceinline__ void gemv_k256_body(const K256Layer& Ly, const int tokens_arg) {
  const int tokens = tokens_arg & (kOutF32Bit - 1);
  const bool out_f32 = (tokens_arg & kOutF32Bit) != 0;
  // The dynamic LDS segment starts at byte 0 (the kernel has no static LDS), so
  // gathers address LDS absolutely; `smem` only sizes the allocation.
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  {
    // absolute LDS addressing is only valid if that holds; folds to nothing when it does
    typedef __attribute__((address_space(3))) unsigned char lds_u8_t;
    if ((uint32_t)(uintptr_t)(lds_u8_t*)smem != 0u) __builtin_trap();
  }

  // ---- row group = blockIdx.x ----
  const int bid = blockIdx.x;
  const int row0 = bid * ROWS;
  if (row0 >= Ly.N) return;  // grid.x is the largest row-group count of the group
  const int G = Ly.G, N = Ly.N, O = Ly.O;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t* const idx_base = Ly.idx;
  const uint16_t* const xp = Ly.x;
  const uint16_t* const sp = Ly.scale;
  const uint16_t* const bp = Ly.wbias;
  const uint16_t* const pp = Ly.perm;
  const size_t row_words = (size_t)Ly.row_words;

  K256_STAMP(kWaves, 0, tid);
  // ---- 1. codebook entry for the LDS image: thread t loads entry t.  Both table
  // pointers are read as scalars; the per-thread choice is an offset, so the load
  // does not wait on a vector fetch of the pointer itself. ----
  const char* const cent0 = (const char*)Ly.cent;
  const ptrdiff_t rdelta = (const char*)Ly.rcent - cent0;
  const u32x4 centry =
      *(const u32x4*)as_global(cent0 + (tid < 256 ? (ptrdiff_t)0 : rdelta) + (size_t)(tid & 255) * 16);

  float acc[TOK][ROWS][8];
  float accb[TOK];
#pragma unroll
  for (int t = 0; t < TOK; ++t) {
    accb[t] = 0.f;
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[t][r][i] = 0.f;
  }

  // gather address bases: replica slot in bits 4..6, residual half of the row: bit 7
  const uint32_t hi = (uint32_t)(lane >> 3) & 1u;
  const uint32_t baseC = ((hi << 3) | (uint32_t)(lane & 7)) << 4;
  const uint32_t baseR = baseC ^ 0x80u;
  const uint32_t selC = 0x0c0c0400u | (hi << 8), selR = 0x0c0c0400u | ((hi ^ 1u) << 8);

  uint32_t pf_word = 0;
  // Straight-line body: no load is predicated.  Lanes past the last column re-read the
  // last 8 columns with x forced to 0; rows past N re-read row N-1 and are not stored.
  for (int base = 0; base < G; base += SW * kSweepCols) {
    // ---- 2. issue every global load of this iteration ----
    u32x4 x_raw[SW][TOK], s_raw[SW], b_raw[SW];
    u32x4 iw[SW][ROWS];
#pragma unroll
    for (int sw = 0; sw < SW; ++sw) {
      const int want = base + sw * kSweepCols + tid * 8;
      const bool valid = want < G;  // G % 8 == 0 (checked on the host)
      const int col0 = valid ? want : G - 8;
      // PERM: sp / bp already hold scale[perm[c]] / bias[perm[c]] (derived state,
      // VptqLayerDesc.scale_permuted); only the activations are gathered.
      u32x4 pv = u32x4{0, 0, 0, 0};
      if (PERM) pv = ld16(pp, (uint32_t)col0 * 2u);
      s_raw[sw] = ld16(sp, (uint32_t)col0 * 2u);
      b_raw[sw] = ld16(bp, (uint32_t)col0 * 2u);
#pragma unroll
      for (int t = 0; t < TOK; ++t) {
        const int te = TOK == 1 ? 0 : (t < tokens ? t : tokens - 1);  // spare token slots repeat the last row
        const u32x4 xv = load8<PERM>(xp + (size_t)te * G, col0, pv);
        const uint32_t keep = valid ? 0xffffffffu : 0u;
        x_raw[sw][t] = u32x4{xv[0] & keep, xv[1] & keep, xv[2] & keep, xv[3] & keep};
      }
#pragma unroll
      for (int r = 0; r < ROWS; ++r) {
        const int row = row0 + r < N ? row0 + r : N - 1;   // wave-uniform
        iw[sw][r] = ld16(idx_base + (size_t)row * row_words, (uint32_t)col0 * 2u);
      }
    }

    // ---- 3. build the LDS codebook image (first iteration only; uniform branch) ----
    if (base == 0) {
      K256_STAMP(kWaves, 1, tid);
      // thread t owns half a bank row: 8 replicas of its entry.  The replica
      // order is rotated by the lane id so the 8 lanes of a ds_write_b128 group
      // hit 8 different 16-byte slots.
      const uint32_t rowp = (tid & 255) * 256 + (tid >> 8) * 128;
#pragma unroll
      for (int q = 0; q < 8; ++q) lds_store16(rowp + (((q + lane) & 7) << 4), centry);
      __syncthreads();
      K256_STAMP(kWaves, 2, tid);
#ifndef VPTQ_K256_TRACE
      // read-ahead for the NEXT launch, issued after the image is built (this
      // workgroup's own loads are in flight ahead of it): one word per 128-byte line of
      // this workgroup's share of the range (the rows it will own if the next layer has
      // this layer's shape -> same XCD L2; otherwise the Infinity Cache still helps).
      // Unconditional, address-clamped load; its value is only "used" in a branch that
      // is never taken, so nothing ever waits for it.
      const long long want = (long long)bid * Ly.pf_chunk + (long long)tid * 128;
      const bool in = tid * 128 < Ly.pf_len && want + 4 <= Ly.pf_bytes;
      const char* pa = in ? Ly.pf + want : (const char*)cent0;
      pf_word = *(const uint32_t*)as_global(pa);
#endif
    }

    // ---- 4. dequantise + accumulate ----
#pragma unroll
    for (int sw = 0; sw < SW; ++sw) {
      float xs[TOK][8];  // FAST only: x * scale in fp32
      if (FAST) {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const float sv = DT::half_of(s_raw[sw][c >> 1], c & 1);
          const float bv = DT::half_of(b_raw[sw][c >> 1], c & 1);
#pragma unroll
          for (int t = 0; t < TOK; ++t) {
            const float xv = DT::half_of(x_raw[sw][t][c >> 1], c & 1);
            // folded form: y = sum (x*s)*(c+r) + sum x*b   (fp32)
            accb[t] = __builtin_fmaf(xv, bv, accb[t]);
            xs[t][c] = xv * sv;
          }
        }
      }
      // ROWS*4 units of one index word (2 elements, 4 gathers); the gathers of
      // unit u+1 are issued before the arithmetic of unit u.
      constexpr int NU = ROWS * 4;
      u32x4 cv[2][2], rv[2][2];
#pragma unroll
      for (int e = 0; e < 2; ++e) gather(iw[sw][0][0], e, baseC, baseR, selC, selR, cv[0][e], rv[0][e]);
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const int r = u >> 2, k = u & 3;
        if (u + 1 < NU) {
#pragma unroll
          for (int e = 0; e < 2; ++e)
            gather(iw[sw][(u + 1) >> 2][(u + 1) & 3], e, baseC, baseR, selC, selR, cv[(u + 1) & 1][e],
                   rv[(u + 1) & 1][e]);
        }
        // stage-major order: the 8 independent pair-chains of the unit advance
        // together, so dependent packed-f16 ops are never back-to-back
        if constexpr (std::is_same<DT, BF16>::value && !FAST) {
          // bf16: the widened roundings from the packed pairs, one scheduled block of four pairs per stage (common.h, round 6)
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            uint32_t w[4] = {cv[u & 1][e][0], cv[u & 1][e][1], cv[u & 1][e][2], cv[u & 1][e][3]};
            const uint32_t rr[4] = {rv[u & 1][e][0], rv[u & 1][e][1], rv[u & 1][e][2], rv[u & 1][e][3]};
            BF16::add4(w, rr);
            BF16::scale_bias4(w, s_raw[sw][k], e, b_raw[sw][k], e);
#pragma unroll
            for (int t = 0; t < TOK; ++t) BF16::fma4(&acc[t][r][0], w, x_raw[sw][t][k], e);
          }
          continue;
        }
        uint32_t w2[2][4];
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
          for (int p = 0; p < 4; ++p) w2[e][p] = DT::add2_g(cv[u & 1][e][p], rv[u & 1][e][p]);
        if (!FAST) {
          // column 2k+e uses half e of pair register k of scale / bias
#pragma unroll
          for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int p = 0; p < 4; ++p) w2[e][p] = DT::mul2_bcast_g(w2[e][p], s_raw[sw][k], e);
#pragma unroll
          for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int p = 0; p < 4; ++p) w2[e][p] = DT::add2_bcast_g(w2[e][p], b_raw[sw][k], e);
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) {
#pragma unroll
          for (int t = 0; t < TOK; ++t) {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
              if (FAST) {
                acc[t][r][2 * p] = DT::fma_lo(w2[e][p], xs[t][2 * k + e], acc[t][r][2 * p]);
                acc[t][r][2 * p + 1] =
                    DT::fma_hi(w2[e][p], xs[t][2 * k + e], acc[t][r][2 * p + 1]);
              } else {
                acc[t][r][2 * p] = DT::fma_lo_h_g(w2[e][p], x_raw[sw][t][k], e, acc[t][r][2 * p]);
                acc[t][r][2 * p + 1] =
                    DT::fma_hi_h_g(w2[e][p], x_raw[sw][t][k], e, acc[t][r][2 * p + 1]);
              }
            }
          }
        }
      }
    }
  }

  K256_STAMP(kWaves, 3, acc[0][0][0] + acc[0][0][1]);
  // ---- 5. reduce over the workgroup's lanes and store ----
  constexpr int kVals = TOK * ROWS * 8;
  constexpr int kStride = kVals + TOK;  // floats per wave in the scratch area
  float* red = (float*)(smem + kScratchOff);  // [kWaves][kStride]
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
    // lane l now holds the wave total of partial l >> kShift
    if ((lane & ((1 << WR::kShift) - 1)) == 0) red[wave * kStride + (lane >> WR::kShift)] = v[0];
  }
  if (FAST) {
#pragma unroll
    for (int t = 0; t < TOK; ++t) {
      const float sum = wave_sum(accb[t]);
      if (lane == 0) red[wave * kStride + kVals + t] = sum;
    }
  }
  K256_STAMP(kWaves, 4, tid);
  __syncthreads();
  K256_STAMP(kWaves, 5, tid);
  if (tid < kVals) {
    const int t = tid / (ROWS * 8), rem = tid - t * (ROWS * 8);
    const int row = row0 + (rem >> 3);
    const int o = row * 8 + (rem & 7);
    if (t < tokens && row < N && o < O) {
      // the tail of the kernel: pairwise sums (3 levels) instead of one chain of 16 adds
      static_assert(kWaves == 8, "final sum");
      float p[kWaves], pb[kWaves];
#pragma unroll
      for (int w = 0; w < kWaves; ++w) {
        p[w] = red[w * kStride + tid];
        pb[w] = FAST ? red[w * kStride + kVals + t] : 0.f;
      }
      float sum = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
      if (FAST) sum += ((pb[0] + pb[1]) + (pb[2] + pb[3])) + ((pb[4] + pb[5]) + (pb[6] + pb[7]));
      if (Ly.bias) sum += DT::to_float(Ly.bias[o]);
      if (out_f32) ((float*)Ly.y)[(size_t)t * O + o] = sum;
      else Ly.y[(size_t)t * O + o] = DT::from_float(sum);
    }
  }
  if (tokens == 0x7fffffff) Ly.y[0] = (uint16_t)pf_word;  // never true: keeps the read-ahead alive
}

// Entry point of grouped launches (and of 2-4 tokens): layer = blockIdx.y.  Every kernel
// argument sits at an offset known at wave start, so all scalar loads go out in one batch (a
// search through the layer table costs one dependent kernarg round trip per step).
template <typename DT, int ROWS, int TOK, int SW, bool PERM, bool FAST>
__global__ __launch_bounds__(kThreads, 4) void gemv_k256_kernel(const K256Params P) {
  int tokens;
  const K256Layer Ly = load_layer_args(tokens);
  gemv_k256_body<DT, ROWS, TOK, SW, PERM, FAST>(Ly, tokens);
}

// Entry point of a single-layer, one-token launch: the pointers and sizes the first loads need
// are 14 dwords of scalar arguments, delivered in SGPRs at wave launch (kernarg preload, see
// gemv_k256m_kernel_1) - no round trip to the kernel-argument segment before the first load
// (~0.3 of the ~5 us of a 4096^2 launch).
template <typename DT, int ROWS, int SW, bool PERM, bool FAST>
__global__ __launch_bounds__(kThreads, 4) void gemv_k256_kernel_1(
    const uint32_t* h_cent, const uint32_t* h_rcent, const uint32_t* h_idx, const uint16_t* h_x,
    const uint16_t* h_scale, const uint16_t* h_wbias, int h_N, int h_G, const K256Params P) {
  K256Layer Ly = P.layer[0];
  Ly.cent = h_cent; Ly.rcent = h_rcent; Ly.idx = h_idx; Ly.x = h_x; Ly.scale = h_scale;
  Ly.wbias = h_wbias; Ly.N = h_N; Ly.G = h_G;
  Ly.row_words = h_G >> 1;  // 16 index bits per column (gemv_k256_eligible)
  Ly.idx = as_global(Ly.idx); Ly.cent = as_global(Ly.cent); Ly.rcent = as_global(Ly.rcent);
  Ly.x = as_global(Ly.x); Ly.y = as_global(Ly.y); Ly.scale = as_global(Ly.scale);
  Ly.wbias = as_global(Ly.wbias); Ly.bias = as_global(Ly.bias); Ly.perm = as_global(Ly.perm);
  Ly.pf = as_global(Ly.pf);
  gemv_k256_body<DT, ROWS, 1, SW, PERM, FAST>(Ly, P.tokens);
}

// ---- host side -------------------------------------------------------------------
bool gemv_k256_eligible(const VptqLayerDesc& d, int tokens) {
  return d.vector_len == 8 && d.num_centroids == 256 && d.num_res_centroids == 256 &&
         d.index_bits == 8 && d.res_bits == 8 && d.num_codebooks == 1 && d.outlier_size == 0 &&
         d.weight_scale != nullptr && d.weight_bias != nullptr && (d.group_size % 8) == 0 &&
         (d.perm == nullptr || (d.scale_permuted != nullptr && d.bias_permuted != nullptr &&
                                (((uintptr_t)d.scale_permuted | (uintptr_t)d.bias_permuted) & 15) == 0)) &&
         d.group_size == d.in_features && d.row_words == d.group_size / 2 && tokens >= 1 &&
         tokens <= 4 && (((uintptr_t)d.indices | (uintptr_t)d.centroids |
                          (uintptr_t)d.res_centroids | (uintptr_t)d.weight_scale |
                          (uintptr_t)d.weight_bias | (uintptr_t)d.perm) & 15) == 0;
}

// ROWS per workgroup: 2 when that still leaves >= 512 workgroups (two per CU) and
// the instantiation stays spill-free under the 128-VGPR budget (f16, one token).
static int pick_rows(int n_rows_total, int tok, bool f16) {
  static std::atomic<int> forced{-1};  // VPTQ_K256_ROWS=1|2: tuning override
  if (forced < 0) { const char* e = vptq::tune_env("VPTQ_K256_ROWS"); forced = e ? atoi(e) : 0; }
  if (forced == 2 && f16 && tok == 1) return 2;
  if (forced == 1) return 1;
  return (f16 && tok == 1 && (n_rows_total + 1) / 2 >= 512) ? 2 : 1;
}

template <typename DT, int ROWS, int TOK, int SW, bool PERM, bool FAST>
static hipError_t launch_inst(const K256Params& P, int grid, hipStream_t st) {
  constexpr int lds = kScratchOff + kWaves * (TOK * ROWS * 8 + TOK) * 4;
  // > 64 KiB of dynamic LDS must be enabled once per device (benign race: idempotent)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  auto allow_lds = [&](const void* kern, std::atomic<bool>& done) {
    if (done) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    done = e == hipSuccess;
    return e;
  };
  if constexpr (TOK == 1) {
    if (P.n_layers == 1) {  // the preloaded-argument entry point
      auto kern = gemv_k256_kernel_1<DT, ROWS, SW, PERM, FAST>;
      static std::atomic<bool> attr_set[64];
      if (hipError_t e = allow_lds((const void*)kern, attr_set[dev]); e != hipSuccess) return e;
      const K256Layer& L0 = P.layer[0];
      hipLaunchKernelGGL(kern, dim3(grid, 1), dim3(kThreads), lds, st, L0.cent, L0.rcent, L0.idx, L0.x,
                         L0.scale, L0.wbias, L0.N, L0.G, P);
      return hipGetLastError();
    }
  }
  auto kern = gemv_k256_kernel<DT, ROWS, TOK, SW, PERM, FAST>;
  static std::atomic<bool> attr_set[64];
  if (hipError_t e = allow_lds((const void*)kern, attr_set[dev]); e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(grid, P.n_layers), dim3(kThreads), lds, st, P);
  return hipGetLastError();
}

template <typename DT, int ROWS, int TOK, bool FAST>
static hipError_t launch_shape(const K256Params& P, int grid, int sw, bool perm, hipStream_t st) {
  if (sw == 1) {
    return perm ? launch_inst<DT, ROWS, TOK, 1, true, FAST>(P, grid, st)
                : launch_inst<DT, ROWS, TOK, 1, false, FAST>(P, grid, st);
  }
  if constexpr (TOK == 4) {
    return hipErrorInvalidValue;  // never chosen (launch_gemv_k256): two sweeps x 4 tokens spill
  } else {
    return perm ? launch_inst<DT, ROWS, TOK, 2, true, FAST>(P, grid, st)
                : launch_inst<DT, ROWS, TOK, 2, false, FAST>(P, grid, st);
  }
}

#define K256_CASE(DT, R, T, F) \
  if (rows == R && tok == T && fast == F) return launch_shape<DT, R, T, F>(P, grid, sw, perm, st);

template <typename DT, bool ALLOW_FAST>
static hipError_t dispatch(const K256Params& P, int grid, int rows, int tok, bool fast, int sw,
                           bool perm, hipStream_t st) {
  K256_CASE(DT, 1, 1, false) K256_CASE(DT, 1, 2, false) K256_CASE(DT, 1, 4, false)
  if constexpr (ALLOW_FAST) {
    K256_CASE(DT, 2, 1, false)
    K256_CASE(DT, 1, 1, true) K256_CASE(DT, 2, 1, true) K256_CASE(DT, 1, 2, true)
  }
  return hipErrorInvalidValue;
}

// Persistent MFMA kernel (gemv_k256m.hip) or the VALU kernel above?  Measured on MI355X
// (profiles/r01/kernel_ab_h*.json, shapes_llama3_*.json): the MFMA kernel wins once the launch
// is large enough - one 8192^2 layer 8.2 vs 10.0 us (exact
// arithmetic 9.4 vs 10.5), 4 x 8192^2 grouped 5.6 vs 7.1 us per layer, 8192x28672 20.6 vs
// 27.4 us, 4096x14336 8.2 vs 11.3 us - and loses below that (one workgroup per CU leaves CUs
// idle: 4096^2 5.3 vs 4.9 us).  The crossover sits where the VALU kernel needs a second round of
// workgroups (more than 512 vector-rows): 8192x5120 (160 row groups) 7.4 vs 9.1 us, 8192x4096
// (128) 7.3 vs 7.0 us.  VPTQ_K256_KERNEL=valu|mfma and the FORCE flags override.
// For bf16 the VALU kernel only has the exact form (arithmetic widened to fp32: ~3x the
// instructions, 23.6 vs 9.4 us per 8192^2 layer); the MFMA kernel's folded form is dtype
// agnostic, so bf16 takes it from 32 row groups on.
struct K256Choice {
  bool mfma;  // persistent MFMA kernel (else the VALU kernel)
  bool fast;  // folded arithmetic (else the reference's per-weight roundings)
  bool sel;   // VPTQ_GEMV_SELECTIVE inside the persistent MFMA kernel: folded + the reference's roundings on the hot blocks
};

static K256Choice choose_kernel(const VptqLayerDesc* descs, int n, int tokens, int flags) {
  static std::atomic<int> forced{-1};
  if (forced < 0) {
    const char* e = vptq::tune_env("VPTQ_K256_KERNEL");
    forced = !e ? 0 : (e[0] == 'v' ? 1 : e[0] == 'm' ? 2 : 0);
  }
  const int tok = tokens > 2 ? 4 : tokens;
  const bool f16 = descs[0].dtype == VPTQ_DTYPE_F16;
  // SELECTIVE without a kernel that implements it = the reference's roundings (EXACT wins when both are set)
  const bool want_sel = (flags & VPTQ_GEMV_SELECTIVE) != 0 && !(flags & VPTQ_GEMV_EXACT);
  bool exact = (flags & VPTQ_GEMV_EXACT) != 0 || want_sel;
  int max_cols = 0;
  bool same_cols = true, perm = false;
  long long row_groups = 0;
  int n_rows[kMaxGroup];
  for (int i = 0; i < n; ++i) {
    max_cols = descs[i].group_size > max_cols ? descs[i].group_size : max_cols;
    // the persistent MFMA kernel is instantiated per column count: one count per launch
    same_cols = same_cols && descs[i].group_size == descs[0].group_size;
    perm = perm || descs[i].perm != nullptr;
    row_groups += gemv_k256m_row_groups(descs[i].num_indices);
    n_rows[i < kMaxGroup ? i : 0] = descs[i].num_indices;
  }
  {
    // selective: where the persistent MFMA kernel would be taken in the folded arithmetic and can carry the corrections
    const long long threshold = f16 ? 144 : 32;
    const bool mfma_ok = same_cols && forced != 1 && !(flags & VPTQ_GEMV_FORCE_VALU) &&
                         (forced == 2 || (flags & VPTQ_GEMV_FORCE_MFMA) || row_groups >= threshold);
    if (want_sel && n <= kMaxGroup && mfma_ok && gemv_k256m_selective_ok(n_rows, n, f16, tok, max_cols, perm))
      return {true, true, true};
  }
  // VALU kernel: its folded-arithmetic instantiations exist for fp16, 1-2 tokens
  K256Choice c = {false, f16 && tok <= 2 && !exact, false};
  if (same_cols && forced != 1 && !(flags & VPTQ_GEMV_FORCE_VALU) &&
      gemv_k256m_supported(tok, f16, !exact, max_cols, perm)) {
    // fp16: from 144 row groups on (where the VALU kernel needs a second round of workgroups;
    // below, its shorter prologue wins - 4096^2, 2 tokens: 6.0 vs 6.9 us).  bf16: the VALU
    // kernel runs widened arithmetic, the MFMA kernel's folded form is dtype agnostic and its exact form rounds on the matrix
    // pipe: from 32 row groups on.  (Round 6 tried 144 for the exact form - on ONE layer replayed from L2 the VALU kernel's
    // smaller row tiles win below that, 4096^2 7.1 vs 7.9 us, 8192 -> 1024 7.0 vs 10.9 - but over HBM-cold weights it is a tie
    // at 4096^2 and the 8B-shaped decode loses 95 us per token: 1806 vs 1711.)
    // (several tokens in the reference's roundings, round 6: 144 for bf16 as well - 4096^2, 2 tokens: 9.1 us against the VALU
    // kernel's 8.3; 8192 x 1024: 13.1 vs 8.1; from 144 row groups on the matrix pipe wins 10 - 50 %, tools/tokens_exact_check.py)
    const long long threshold = f16 || (exact && tok > 1) ? 144 : 32;
    if (forced == 2 || (flags & VPTQ_GEMV_FORCE_MFMA) || row_groups >= threshold)
      c = {true, !exact, false};
  }
  return c;
}

static const char* choice_name(K256Choice c) {
  if (c.sel) return "gemv_k256m_kernel<selective>";
  if (c.mfma) return c.fast ? "gemv_k256m_kernel<fast>" : "gemv_k256m_kernel";
  return c.fast ? "gemv_k256_kernel<fast>" : "gemv_k256_kernel";
}

const char* gemv_k256_name(const VptqLayerDesc& d, int tokens, int flags) {
  return choice_name(choose_kernel(&d, 1, tokens, flags));
}

// which kernel one grouped launch of these layers uses
const char* gemv_k256_group_name(const VptqLayerDesc* descs, int n, int tokens, int flags) {
  return choice_name(choose_kernel(descs, n, tokens, flags));
}

// A grouped launch of the persistent kernel lasts as long as its slowest workgroup: max over the layers of (row groups per
// workgroup) row-group times + one prologue.  Three equal 8192^2 layers on 256 CUs are 768 row groups on 3 x 85 workgroups - four
// per workgroup where three would do - and came out SLOWER than three launches (10.7 vs 9.5 us per layer, round 6,
// profiles/r06/chain_vs_group.txt).  For 2 - 4 layers every split into launches is priced - sum over the launches of (units + rho),
// rho = the fixed cost of a launch in row-group times (2.8 us against 0.76 / 0.48 / 1.09 ns per column of a row group: fp16
// reference roundings / folded / bf16 reference roundings; calibrated on 8192^2 single and grouped launches) - and the cheapest is
// issued.  part[i] = launch of layer i, numbered from 0 in order of first appearance.
static int best_split(const VptqLayerDesc* descs, int n, K256Choice c, int* part) {
  for (int i = 0; i < n; ++i) part[i] = 0;
  if (!c.mfma || n < 2 || n > 4) return 1;
  const bool f16 = descs[0].dtype == VPTQ_DTYPE_F16;
  const double ns_per_col = c.fast ? 0.48 : f16 ? 0.76 : 1.09;
  const double rho = 2800.0 / (ns_per_col * descs[0].group_size);
  int rows[4];
  for (int i = 0; i < n; ++i) rows[i] = descs[i].num_indices;
  double best = 1e30;
  int best_parts = 1, cur[4] = {0, 0, 0, 0};
  // restricted growth strings: cur[0] = 0, cur[i] <= max(cur[0..i-1]) + 1
  for (;;) {
    int parts = 0;
    for (int i = 0; i < n; ++i) parts = cur[i] + 1 > parts ? cur[i] + 1 : parts;
    double cost = 0;
    for (int q = 0; q < parts; ++q) {
      int r[4], m = 0;
      for (int i = 0; i < n; ++i) if (cur[i] == q) r[m++] = rows[i];
      cost += gemv_k256m_launch_units(r, m) + rho;
    }
    if (cost < best - 1e-9) { best = cost; best_parts = parts; for (int i = 0; i < n; ++i) part[i] = cur[i]; }
    int i = n - 1;
    for (; i > 0; --i) {
      int mx = 0;
      for (int j = 0; j < i; ++j) mx = cur[j] > mx ? cur[j] : mx;
      if (cur[i] <= mx) { ++cur[i]; break; }
      cur[i] = 0;
    }
    if (i == 0) break;
  }
  return best_parts;
}

static hipError_t launch_gemv_k256_one(const VptqLayerDesc* descs, int n, const void* const* x, void* const* y, int tokens, int flags,
                                       hipStream_t st, bool may_split);

hipError_t launch_gemv_k256(const VptqLayerDesc* descs, int n, const void* const* x,
                            void* const* y, int tokens, int flags, hipStream_t st) {
  return launch_gemv_k256_one(descs, n, x, y, tokens, flags, st, true);
}

static hipError_t launch_gemv_k256_one(const VptqLayerDesc* descs, int n, const void* const* x, void* const* y, int tokens, int flags,
                                       hipStream_t st, bool may_split) {
  if (n < 1 || n > kMaxGroup) return hipErrorInvalidValue;
  K256Params P;
  P.n_layers = n;
  P.tokens = tokens | ((flags & VPTQ_GEMV_OUT_F32) ? kOutF32Bit : 0);
  int total_rows = 0;
  for (int i = 0; i < n; ++i) total_rows += descs[i].num_indices;
  const int tok = tokens > 2 ? 4 : tokens;
  const bool f16 = descs[0].dtype == VPTQ_DTYPE_F16;
  const K256Choice choice = choose_kernel(descs, n, tokens, flags);
  const bool mfma = choice.mfma, fast = choice.fast;
  if (may_split && tokens == 1) {
    int part[4];
    const int parts = best_split(descs, n, choice, part);
    if (parts > 1) {
      // the same kernel and arithmetic as the group would have run (a part alone might fall under the row-group threshold)
      const int sub_flags = flags | VPTQ_GEMV_FORCE_MFMA;
      for (int q = 0; q < parts; ++q) {
        VptqLayerDesc d[4];
        const void* xs[4];
        void* ys[4];
        int m = 0;
        for (int i = 0; i < n; ++i)
          if (part[i] == q) { d[m] = descs[i]; xs[m] = x[i]; ys[m] = y[i]; ++m; }
        if (hipError_t e = launch_gemv_k256_one(d, m, xs, ys, tokens, sub_flags, st, false); e != hipSuccess) return e;
      }
      return hipSuccess;
    }
  }
  int maxG = 0;
  bool perm = false;
  for (int i = 0; i < n; ++i) {
    maxG = descs[i].group_size > maxG ? descs[i].group_size : maxG;
    perm = perm || descs[i].perm != nullptr;
  }
  const int rows = mfma ? kMRows : pick_rows(total_rows, tok, f16);
  const int wg_threads = mfma ? 1024 : kThreads;
  int grid = 0;
  for (int i = 0; i < n; ++i) {
    const VptqLayerDesc& d = descs[i];
    K256Layer& Ly = P.layer[i];
    Ly.idx = (const uint32_t*)d.indices;
    Ly.cent = (const uint32_t*)d.centroids;
    Ly.rcent = (const uint32_t*)d.res_centroids;
    Ly.x = (const uint16_t*)x[i];
    Ly.y = (uint16_t*)y[i];
    Ly.scale = (const uint16_t*)(d.perm ? d.scale_permuted : d.weight_scale);
    Ly.wbias = (const uint16_t*)(d.perm ? d.bias_permuted : d.weight_bias);
    Ly.bias = (const uint16_t*)d.bias;
    Ly.perm = d.perm;
    Ly.N = d.num_indices;
    Ly.G = d.group_size;
    Ly.O = d.out_features;
    Ly.row_words = d.row_words;
    const int n_wg = (d.num_indices + rows - 1) / rows;
    Ly.wgs = 0;
    Ly.pf = (const char*)d.prefetch;
    Ly.pf_bytes = d.prefetch ? d.prefetch_bytes : 0;
    long long chunk = d.prefetch ? (d.prefetch_bytes + n_wg - 1) / n_wg : 0;
    chunk = (chunk + 127) / 128 * 128;
    Ly.pf_chunk = (int)chunk;
    static std::atomic<int> pf_cap{-1};  // VPTQ_PF_BYTES: cap on the bytes each workgroup reads ahead
    if (pf_cap < 0) { const char* e = vptq::tune_env("VPTQ_PF_BYTES"); pf_cap = e ? atoi(e) : 1 << 30; }
    const long long cap = pf_cap.load();
    long long len = chunk < cap ? chunk : cap;
    if (len > wg_threads * 128) len = wg_threads * 128;  // one line per thread
    Ly.pf_len = (int)len;
    Ly.slots = 0;
    grid = n_wg > grid ? n_wg : grid;  // grid.x; grid.y = layer
  }
  // all layers of a group share one instantiation: widest column count decides the
  // sweeps per iteration, any permutation selects the gather variant (grouped layers
  // must agree on it, checked by the caller)
  // two sweeps per iteration keep more loads in flight, but the 4-token instantiation
  // only stays spill-free with one
  if (mfma) return launch_gemv_k256m(P, tok, f16, fast, maxG, perm, st, choice.sel);
  const int sw = (maxG > kSweepCols && tok != 4) ? 2 : 1;
  return f16 ? dispatch<F16, true>(P, grid, rows, tok, fast, sw, perm, st)
             : dispatch<BF16, false>(P, grid, rows, tok, false, sw, perm, st);
}

}  // namespace vptq
