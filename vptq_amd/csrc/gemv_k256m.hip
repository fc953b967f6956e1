// Fused dequant + GEMV for the canonical "2-bit" VPTQ format (v = 8, 256 + 256 centroids),
// MFMA-accumulate variant of gemv_k256.hip.  Same contract, same reference
// (csrc/kernels/quant_gemv.cuh:11-186, csrc/quant_gemv.cu:203-235).
//
// Why a second kernel: gemv_k256.hip is bound by VALU issue (22 instructions per index in
// the exact form, 14 folded; tools/ubench.hip) and by 2-way LDS bank conflicts of its
// 8-replica codebook image (8.8 LDS cycles per gather, tools/ubench_lds.hip).  Here
//  * the multiply-accumulate moves to the matrix pipe.  A GEMV is no contraction for MFMA,
//    but v_mfma_f32_4x4x4_16b_f16 computes 16 independent 4x4 blocks D = X * W + D per
//    instruction; with X = x' * I (lane i of a block supplies x' * e_i) and W = the four
//    gathered centroid halves of the block's four lanes (lane j supplies ITS four weights as
//    column j), D[i][j] = x' * W_j[i]: every lane accumulates x' times its own four weights
//    in fp32, i.e. four v_fma_mix_f32 become one MFMA that issues beside the VALU stream.
//    The four lanes of a block must share x', so a block is ONE input column of FOUR
//    consecutive vector-rows: lane = (block b = column, j = row).
//  * folded form (VPTQ_GEMV_FAST_MATH): y = sum_g (c + r) * f16(s_g x_g) + sum_g b_g x_g; the
//    main and residual halves are separate MFMAs into the same accumulator, so c + r is never
//    formed: 4 v_perm_b32 + 4 MFMA per index instead of 14 VALU.
//  * exact form: the reference's three roundings r16(r16(r16(c+r)*s)+b) stay packed-f16 VALU
//    (12 ops), the 8 fp32 FMAs become 2 MFMAs (f16 x f16 products are exact in fp32, so the
//    sum is the same fma chain): 16 VALU + 2 MFMA instead of 22 VALU.
//  * 1024-thread workgroups, one per CU (4 waves / SIMD), own 4 vector-rows (32 outputs) over
//    all input columns; the LDS image holds 16 replicas of both codebooks (128 KiB, address
//    = table << 16 | entry << 8 | replica << 4, replica = lane & 15): every lane of a
//    ds_read_b128 group reads its own bank quad -> conflict free (5.1 LDS cycles per gather).
#include "common.h"
#include "kernels.h"
#include "k256.h"

namespace vptq {

constexpr int kMThreads = 1024;
constexpr int kMWaves = kMThreads / 64;
constexpr int kMSweepCols = kMWaves * 16 * 8;  // 16 blocks (columns chunks of 8) per wave
constexpr int kMTableBytes = 131072;
constexpr int kMMaxLds = 163840;          // 160 KiB per CU
constexpr int kMMaxCols = 14336;          // staged activations must fit beside the image

static __device__ __forceinline__ u32x4 ldg16(const void* base, uint32_t byte_off) {
  return *(const u32x4*)as_global((const char*)base + byte_off);
}

template <bool PERM>
static __device__ __forceinline__ u32x4 ldg8x16(const uint16_t* __restrict__ p, int col0,
                                                const u32x4& pv) {
  if (!PERM) return ldg16(p, (uint32_t)col0 * 2u);
  const uint16_t* const g = as_global(p);
  u32x4 r;
#pragma unroll
  for (int q = 0; q < 4; ++q)
    r[q] = (uint32_t)g[pv[q] & 0xffffu] | ((uint32_t)g[pv[q] >> 16] << 16);
  return r;
}

template <typename DT, int TOK, int SW, int NST, bool PERM, bool FAST>
__global__ __launch_bounds__(kMThreads) void gemv_k256m_kernel(const K256Params P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  {
    typedef __attribute__((address_space(3))) unsigned char lds_u8_t;
    if ((uint32_t)(uintptr_t)(lds_u8_t*)smem != 0u) __builtin_trap();  // absolute LDS addressing
  }

  // layer = blockIdx.y, row group = blockIdx.x: every kernel argument this workgroup needs
  // sits at an offset known at wave start, so all scalar loads go out in ONE batch (a
  // search through the layer table costs one dependent kernarg round trip per step, and
  // the first vector load cannot be issued before the last one returns)
  int tokens;
  const K256Layer Ly = load_layer_args(tokens);
  const int bid = blockIdx.x;
  const int row0 = bid * kMRows;
  if (row0 >= Ly.N) return;  // grid.x is the largest row-group count of the group
  const int G = Ly.G, N = Ly.N, O = Ly.O;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 3;     // vector-row inside the group = position inside the MFMA block
  const int blk = lane >> 2;  // MFMA block = chunk of 8 columns
  const uint16_t* const xp = Ly.x;
  const uint16_t* const sp = Ly.scale;
  const uint16_t* const bp = Ly.wbias;
  const uint16_t* const pp = Ly.perm;
  const uint32_t row_bytes = (uint32_t)Ly.row_words * 4u;
  const char* const idx_row0 = (const char*)Ly.idx + (size_t)row0 * row_bytes;  // wave-uniform
  const uint32_t my_row_off = (uint32_t)(row0 + j < N ? j : N - 1 - row0) * row_bytes;

  K256_STAMP(kMWaves, 0, tid);
  // ---- 1. LDS codebook image first: thread t replicates entry (t >> 1) & 255 of table t >> 9
  // into 8 of its 16 slots (rotated by the lane id: the 8 lanes of a ds_write_b128 group hit
  // 8 different slots).  Nothing else is requested before these stores are issued: a wave
  // issues in order, and behind a backed-up vector-memory queue the image - which all 16
  // waves wait for - was built 1.5-3 us later (tools/trace_k256m.py).  An LDS-DMA fill
  // (global_load_lds_dwordx4) was slower still.
  {
    const char* const c0 = (const char*)Ly.cent;
    const ptrdiff_t rdelta = (const char*)Ly.rcent - c0;
    const u32x4 centry = *(const u32x4*)as_global(c0 + (tid < 512 ? (ptrdiff_t)0 : rdelta) +
                                                  (size_t)((tid >> 1) & 255) * 16);
    const uint32_t rowp = ((uint32_t)(tid >> 9) << 16) | ((uint32_t)((tid >> 1) & 255) << 8) |
                          ((uint32_t)(tid & 1) << 7);
#pragma unroll
    for (int q = 0; q < 8; ++q) lds_store16(rowp + (((q + lane) & 7) << 4), centry);
  }
  __builtin_amdgcn_sched_barrier(0);
  const char* const cent0 = (const char*)Ly.cent;

  f32x4 acc[TOK][2];
  float accb[TOK];
#pragma unroll
  for (int t = 0; t < TOK; ++t) {
    accb[t] = 0.f;
    acc[t][0] = f32x4{0.f, 0.f, 0.f, 0.f};
    acc[t][1] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // gather address = perm(word, base, sel): {0, base.b2 = table, index byte, base.b0 = replica}
  const uint32_t baseC = (uint32_t)(lane & 15) << 4;
  const uint32_t baseR = baseC | 0x10000u;
  // x operand of the MFMA: x' * e_j as two packed pairs, cut out of a packed x' register
  // (column parity h picks the half) by one v_perm_b32 with a per-lane selector
  const uint32_t selA[2] = {j == 0 ? 0x0c0c0504u : j == 1 ? 0x05040c0cu : 0x0c0c0c0cu,
                            j == 0 ? 0x0c0c0706u : j == 1 ? 0x07060c0cu : 0x0c0c0c0cu};
  const uint32_t selB[2] = {j == 2 ? 0x0c0c0504u : j == 3 ? 0x05040c0cu : 0x0c0c0c0cu,
                            j == 2 ? 0x0c0c0706u : j == 3 ? 0x07060c0cu : 0x0c0c0c0cu};

  // LDS map: [0, 128 KiB) codebook image | activations staged once per workgroup, one array
  // of G + 8 values per token (the last 8 are zeros: the operand of columns past G) | scratch
  const uint32_t xs_bytes = (uint32_t)G * 2u + 32u;  // + 8 zeros + a 16-byte dump slot
  const uint32_t xs_off = kMTableBytes;
  const uint32_t red_off = xs_off + TOK * xs_bytes;

  // ---- 1. global loads: codebook entry (above), the activations this thread stages, then
  // the first index words.  None is predicated (clamped addresses). ----
  // staging: thread t owns columns 8t.. of each 8192-column block.  FAST stages f16(s * x)
  // and sums b * x; the exact form stages x itself.
  constexpr int kStageCols = kMThreads * 8;
  // NST = blocks of 8192 columns (1 or 2).  Straight-line: no load or store is predicated,
  // so the compiler can wait for exactly the loads a store needs.
  u32x4 st_x[NST][TOK], st_s[NST], st_b[NST];
#pragma unroll
  for (int k = 0; k < NST; ++k) {
    {
      const int want = k * kStageCols + tid * 8;
      const int col0 = want < G ? want : G - 8;
      u32x4 pv = u32x4{0, 0, 0, 0};
      if (PERM) pv = ldg16(pp, (uint32_t)col0 * 2u);
      if (FAST) {
        st_s[k] = ldg16(sp, (uint32_t)col0 * 2u);
        st_b[k] = ldg16(bp, (uint32_t)col0 * 2u);
      }
#pragma unroll
      for (int t = 0; t < TOK; ++t) {
        const int te = t < tokens ? t : tokens - 1;
        st_x[k][t] = ldg8x16<PERM>(xp + (size_t)te * G, col0, pv);
      }
    }
  }

  uint32_t pf_word = 0;
  // ---- 2. index words: a rotating queue of SW sweeps in flight (one sweep = 2048 columns
  // of the 4 rows = 16 bytes per lane).  Sweep s + SW is requested when sweep s is consumed.
  constexpr int kMaxSweeps = kMMaxCols / kMSweepCols;  // 7
  const int n_sweeps = (G + kMSweepCols - 1) / kMSweepCols;
  u32x4 s_raw[SW], b_raw[SW], iw[SW];
  auto issue_sweep = [&](int s) {
    const int want = s * kMSweepCols + (wave * 16 + blk) * 8;
    const int col0 = want < G ? want : G - 8;  // G % 8 == 0 (host check)
    if (!FAST) {
      s_raw[s % SW] = ldg16(sp, (uint32_t)col0 * 2u);
      b_raw[s % SW] = ldg16(bp, (uint32_t)col0 * 2u);
    }
    iw[s % SW] = ldg16(idx_row0, my_row_off + (uint32_t)col0 * 2u);
  };

  // every sweep of the queue is requested now; the staged activations are written when
  // their loads return (in-order vmcnt: the image, requested earlier, has landed by then)
  // (unconditional: a sweep past the last column re-reads the last 8 columns - with a
  // conditional load in between the compiler can only wait for ALL loads before the stores)
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int s = 0; s < SW; ++s) issue_sweep(s);
  __builtin_amdgcn_sched_barrier(0);
  K256_STAMP(kMWaves, 1, tid);
  {
    // s_waitcnt vmcnt(K), K = vector loads issued above for the queue: everything older -
    // the activations to stage - has landed (vmcnt retires in order)
    constexpr int K = FAST ? SW : 3 * SW;
    __builtin_amdgcn_s_waitcnt((K & 15) | (7 << 4) | (15 << 8) | ((K >> 4) << 14));
  }
  {
#pragma unroll
    for (int k = 0; k < NST; ++k) {
      const int want = k * kStageCols + tid * 8;
      const bool valid = want < G;
      // columns past G: x = 0 (adds nothing to sum b * x) and the store goes to the dump slot
      const uint32_t keep = valid ? 0xffffffffu : 0u;
      const uint32_t dst = xs_off + (uint32_t)(valid ? want : G + 8) * 2u;
#pragma unroll
      for (int t = 0; t < TOK; ++t) {
        u32x4 v = st_x[k][t];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] &= keep;
        if (FAST) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            accb[t] = DT::dot2(v[q], st_b[k][q], accb[t]);  // sum b * x
            v[q] = DT::mul2(v[q], st_s[k][q]);              // f16(s * x)
          }
        }
        lds_store16(dst + t * xs_bytes, v);
      }
    }
    if (tid < TOK) lds_store16(xs_off + tid * xs_bytes + (uint32_t)G * 2u, u32x4{0, 0, 0, 0});
  }
  __syncthreads();
  K256_STAMP(kMWaves, 2, tid);
#ifndef VPTQ_K256_TRACE
  {
    // read-ahead of the next launch's index stream (see gemv_k256.hip)
    const long long want = (long long)bid * Ly.pf_chunk + (long long)tid * 128;
    const bool in = tid * 128 < Ly.pf_len && want + 4 <= Ly.pf_bytes;
    const char* pa = in ? Ly.pf + want : (const char*)cent0;
    pf_word = *(const uint32_t*)as_global(pa);
  }
#endif

  // ---- 3. dequantise + accumulate: units of one index (2 gathers), kAhead units ahead ----
#pragma unroll
  for (int s = 0; s < kMaxSweeps; ++s) {
    if (s < n_sweeps) {  // wave-uniform
      const int want = s * kMSweepCols + (wave * 16 + blk) * 8;
      const uint32_t xaddr = xs_off + (uint32_t)(want < G ? want : G) * 2u;  // past G: zeros
      u32x4 xq[TOK];
#pragma unroll
      for (int t = 0; t < TOK; ++t) xq[t] = lds_load16(xaddr + t * xs_bytes);
      const u32x4 words = iw[s % SW];
      const u32x4 sv = FAST ? u32x4{0, 0, 0, 0} : s_raw[s % SW];
      const u32x4 bv = FAST ? u32x4{0, 0, 0, 0} : b_raw[s % SW];

      constexpr int NU = 8;
      constexpr int kAhead = 3;
      u32x4 cv[kAhead + 1], rv[kAhead + 1];
      auto gather = [&](int u) {
        const uint32_t w = words[u >> 1];
        const int h = u & 1;
        const uint32_t aC = __builtin_amdgcn_perm(w, baseC, h ? 0x0c020600u : 0x0c020400u);
        const uint32_t aR = __builtin_amdgcn_perm(w, baseR, h ? 0x0c020700u : 0x0c020500u);
        cv[u % (kAhead + 1)] = lds_load16(aC);
        rv[u % (kAhead + 1)] = lds_load16(aR);
      };
#pragma unroll
      for (int u = 0; u < kAhead; ++u) gather(u);
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        if (u + kAhead < NU) gather(u + kAhead);
        const int q = u >> 1, h = u & 1;
        const u32x4 c = cv[u % (kAhead + 1)], r = rv[u % (kAhead + 1)];
        if (FAST) {
#pragma unroll
          for (int t = 0; t < TOK; ++t) {
            const u32x2 xo = {__builtin_amdgcn_perm(xq[t][q], 0u, selA[h]),
                              __builtin_amdgcn_perm(xq[t][q], 0u, selB[h])};
            acc[t][0] = DT::mfma4(xo, u32x2{c[0], c[1]}, acc[t][0]);
            acc[t][1] = DT::mfma4(xo, u32x2{c[2], c[3]}, acc[t][1]);
            acc[t][0] = DT::mfma4(xo, u32x2{r[0], r[1]}, acc[t][0]);
            acc[t][1] = DT::mfma4(xo, u32x2{r[2], r[3]}, acc[t][1]);
          }
        } else {
          uint32_t w2[4];
#pragma unroll
          for (int p = 0; p < 4; ++p) w2[p] = DT::add2(c[p], r[p]);
#pragma unroll
          for (int p = 0; p < 4; ++p) w2[p] = DT::mul2_bcast(w2[p], sv[q], h);
#pragma unroll
          for (int p = 0; p < 4; ++p) w2[p] = DT::add2_bcast(w2[p], bv[q], h);
#pragma unroll
          for (int t = 0; t < TOK; ++t) {
            const u32x2 xo = {__builtin_amdgcn_perm(xq[t][q], 0u, selA[h]),
                              __builtin_amdgcn_perm(xq[t][q], 0u, selB[h])};
            acc[t][0] = DT::mfma4(xo, u32x2{w2[0], w2[1]}, acc[t][0]);
            acc[t][1] = DT::mfma4(xo, u32x2{w2[2], w2[3]}, acc[t][1]);
          }
        }
      }
      if (s + SW < kMaxSweeps && s + SW < n_sweeps) issue_sweep(s + SW);
    }
  }

  K256_STAMP(kMWaves, 3, acc[0][0][0] + acc[0][1][0]);
  // ---- reduce over the 16 column blocks of the wave, then over the 16 waves ----
  // lane (blk, j) holds 8 partial outputs (t = 0..7) of vector-row j.  Lane bits 5 and 4 by
  // swap-and-add (halving the values carried), bits 3 and 2 by DPP row rotations, which keep
  // lane & 3: afterwards lane l holds outputs 4*bit5 + 2*bit4 + {0, 1} of row l & 3.
  constexpr int kStride = TOK * 32 + TOK;
  float* red = (float*)(smem + red_off);  // [kMWaves][kStride]
#pragma unroll
  for (int t = 0; t < TOK; ++t) {
    float v[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] = acc[t][0][i]; v[4 + i] = acc[t][1][i]; }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[i]), __float_as_uint(v[i + 4]),
                                                false, false);
      v[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[i]), __float_as_uint(v[i + 2]),
                                                false, false);
      v[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) v[i] = row_ror_add<4>(row_ror_add<8>(v[i]));
    if ((lane & 12) == 0) {
      const int o8 = ((lane >> 5) & 1) * 4 + ((lane >> 4) & 1) * 2;
      red[wave * kStride + t * 32 + j * 8 + o8] = v[0];
      red[wave * kStride + t * 32 + j * 8 + o8 + 1] = v[1];
    }
    if (FAST) {
      const float sum = wave_sum(accb[t]);
      if (lane == 0) red[wave * kStride + TOK * 32 + t] = sum;
    }
  }
  K256_STAMP(kMWaves, 4, tid);
  __syncthreads();
  K256_STAMP(kMWaves, 5, tid);
  if (tid < TOK * 32) {
    const int t = tid >> 5, rem = tid & 31;
    const int row = row0 + (rem >> 3);
    const int o = row * 8 + (rem & 7);
    if (t < tokens && row < N && o < O) {
      float sum = 0.f;
#pragma unroll
      for (int w = 0; w < kMWaves; ++w) {
        sum += red[w * kStride + tid];
        if (FAST) sum += red[w * kStride + TOK * 32 + t];
      }
      if (Ly.bias) sum += DT::to_float(Ly.bias[o]);
      Ly.y[(size_t)t * O + o] = DT::from_float(sum);
    }
  }
  if (tokens == 0x7fffffff) Ly.y[0] = (uint16_t)pf_word;  // never true: keeps the read-ahead alive
}

// ---- host side -------------------------------------------------------------------
template <typename DT, int TOK, int SW, int NST, bool PERM, bool FAST>
static hipError_t launch_m(const K256Params& P, int grid, int max_cols, hipStream_t st) {
  auto kern = gemv_k256m_kernel<DT, TOK, SW, NST, PERM, FAST>;
  const int lds = kMTableBytes + TOK * (max_cols * 2 + 32) + kMWaves * (TOK * 32 + TOK) * 4;
  if (lds > kMMaxLds) return hipErrorInvalidValue;
  static bool attr_set[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (!attr_set[dev]) {
    hipError_t e = hipFuncSetAttribute((const void*)kern,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, kMMaxLds);
    if (e != hipSuccess) return e;
    attr_set[dev] = true;
  }
  hipLaunchKernelGGL(kern, dim3(grid, P.n_layers), dim3(kMThreads), lds, st, P);
  return hipGetLastError();
}

template <typename DT, int TOK, bool FAST>
static hipError_t launch_m_shape(const K256Params& P, int grid, int sw, bool perm, int max_cols,
                                 hipStream_t st) {
#define K256M_CASE(S, N, PM) \
  if (sw == S && nst == N && perm == PM) return launch_m<DT, TOK, S, N, PM, FAST>(P, grid, max_cols, st);
  const int nst = max_cols > kMThreads * 8 ? 2 : 1;
  K256M_CASE(1, 1, false) K256M_CASE(1, 1, true)
  K256M_CASE(2, 1, false) K256M_CASE(2, 1, true)
  K256M_CASE(4, 1, false) K256M_CASE(4, 1, true)
  K256M_CASE(4, 2, false) K256M_CASE(4, 2, true)
#undef K256M_CASE
  return hipErrorInvalidValue;
}

bool gemv_k256m_supported(int tok, bool f16, bool fast, int max_cols) {
  (void)fast;
  return f16 && tok == 1 && max_cols <= kMMaxCols;
}

hipError_t launch_gemv_k256m(const K256Params& P, int grid, int tok, bool f16, bool fast,
                             int max_cols, bool perm, hipStream_t st) {
  if (!gemv_k256m_supported(tok, f16, fast, max_cols)) return hipErrorInvalidValue;
  static int forced_sw = -1;  // VPTQ_K256M_SW=1|2|4: tuning override
  if (forced_sw < 0) { const char* e = getenv("VPTQ_K256M_SW"); forced_sw = e ? atoi(e) : 0; }
  // sweeps of index words in flight per lane (the queue depth, not the column count)
  int sw = max_cols > 2 * kMSweepCols ? 4 : max_cols > kMSweepCols ? 2 : 1;
  if ((forced_sw == 1 || forced_sw == 2 || forced_sw == 4) && max_cols <= kMThreads * 8) sw = forced_sw;
  return fast ? launch_m_shape<F16, 1, true>(P, grid, sw, perm, max_cols, st)
              : launch_m_shape<F16, 1, false>(P, grid, sw, perm, max_cols, st);
}

}  // namespace vptq
