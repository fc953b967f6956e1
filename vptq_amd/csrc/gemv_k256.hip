// Fused dequant + GEMV for the canonical "2-bit" VPTQ format on MI355X (gfx950):
//   v = 8, k = 256 main + 256 residual centroids (T = 16 bits / index pair),
//   one codebook, no outlier columns, weight_scale/weight_bias present.
//
// Replaces WqA16WithOutliers_PackIndice + tmp.sum(-1)
// (reference csrc/kernels/quant_gemv.cuh:11-186, csrc/quant_gemv.cu:203-235).
//
// Design:
//  * one 512-thread workgroup per CU owns ROWS complete vector-rows (8*ROWS
//    outputs) over ALL input columns -> no split-K buffer, no second kernel;
//  * every lane streams 16-byte pieces of the packed index rows (8 column
//    indices per load, 1 KiB per wave instruction, fully coalesced); all loads
//    of an iteration are issued before the first use so the whole slab of a
//    workgroup is in flight at once;
//  * both codebooks live in LDS, REPLICATED 16x in a bank-partitioned image
//    (entry e, replica q at byte e*256 + q*16; lane l always reads replica
//    l & 15): the two ds_read_b128 gathers per index are bank-conflict-free by
//    construction, for any index pattern (a plain 4-KiB table costs ~3x);
//  * the LDS address of a gather is ONE v_perm_b32 (index byte -> bits 8..15,
//    lane slot -> bits 4..7, table -> bit 16);
//  * weights are rebuilt with the reference CPU path's roundings
//    (r16(r16(r16(c+r)*s)+b), packed fp16 VALU) and x*w accumulates in fp32.
#include "common.h"
#include "kernels.h"

namespace vptq {

constexpr int kThreads = 512;
constexpr int kWaves = kThreads / 64;
constexpr int kSweepCols = kThreads * 8;  // columns covered by one sweep of the WG
constexpr int kSW = 2;                    // sweeps issued back-to-back per iteration
constexpr int kMaxGroup = 32;

struct K256Layer {
  const uint32_t* idx;    // [N, row_words]
  const uint32_t* cent;   // [256, 8] as 4 dwords per entry
  const uint32_t* rcent;  // [256, 8]
  const uint16_t* x;      // [tokens, I]
  uint16_t* y;            // [tokens, O]
  const uint16_t* scale;  // [I]
  const uint16_t* wbias;  // [I]
  const uint16_t* bias;   // [O] or null
  const uint16_t* perm;   // [I] or null
  int N, G, O, row_words;
  int wg_begin;  // first workgroup id of this layer
  int pad_;
};

struct K256Params {
  int n_layers;
  int tokens;
  K256Layer layer[kMaxGroup];
};

// LDS image ---------------------------------------------------------------------
// TAB == 1: [2 tables][256 entries][16 replicas][16 B] = 128 KiB, then scratch.
// TAB == 0: [2 tables][256 entries][16 B]               =   8 KiB, then scratch.
template <int TAB>
struct Lds {
  static constexpr int kTableBytes = TAB ? 65536 : 4096;
  static constexpr int kScratchOff = 2 * kTableBytes;
};

// one element: gather both codebook entries (two ds_read_b128)
template <int TAB>
static __device__ __forceinline__ void gather(uint32_t w, int h, uint32_t baseC, uint32_t baseR,
                                              u32x4& cv, u32x4& rv) {
  uint32_t aC, aR;
  if (TAB == 1) {
    // D = {0, base.b2, w.byte, base.b0}: (index << 8) | lane slot | table bit
    aC = __builtin_amdgcn_perm(w, baseC, h ? 0x0c020600u : 0x0c020400u);
    aR = __builtin_amdgcn_perm(w, baseR, h ? 0x0c020700u : 0x0c020500u);
  } else {
    aC = ((w >> (16 * h)) & 0xffu) << 4;
    aR = (((w >> (16 * h + 8)) & 0xffu) << 4) + Lds<0>::kTableBytes;
  }
  cv = lds_load16(aC);
  rv = lds_load16(aR);
}

template <typename DT, int ROWS, int TOK, bool FAST, int TAB, int RED>
__global__ __launch_bounds__(kThreads) void gemv_k256_kernel(const K256Params P) {
  // The dynamic LDS segment starts at byte 0 (the kernel has no static LDS), so
  // gathers address LDS absolutely; `smem` is only used to size the allocation.
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  using L = Lds<TAB>;

  // ---- which layer / row group is this workgroup? (wave-uniform) ----
  const int bid = blockIdx.x;
  int li = 0;
  for (int l = 1; l < P.n_layers; ++l)
    if (bid >= P.layer[l].wg_begin) li = l;
  const K256Layer& Ly = P.layer[li];
  const int tokens = P.tokens;
  const int row0 = (bid - Ly.wg_begin) * ROWS;
  const int G = Ly.G, N = Ly.N, O = Ly.O;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  // ---- 1. codebook entry for the LDS image: thread t loads entry t ----
  const uint32_t* csrc = (tid < 256 ? Ly.cent : Ly.rcent) + (tid & 255) * 4;
  const u32x4 centry = *(const u32x4*)csrc;

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

  // gather address bases: lane slot in bits 4..7, table select in bit 16
  const uint32_t baseC = (uint32_t)(lane & 15) << 4;
  const uint32_t baseR = baseC | (uint32_t)L::kTableBytes;

  bool tables_ready = false;

  for (int base = 0; base < G; base += kSW * kSweepCols) {
    // ---- 2. issue every global load of this iteration ----
    u32x4 xs_raw[kSW][TOK], s_raw[kSW], b_raw[kSW];
    u32x4 iw[kSW][ROWS];
    bool valid[kSW];
#pragma unroll
    for (int sw = 0; sw < kSW; ++sw) {
      const int col0 = base + sw * kSweepCols + tid * 8;
      valid[sw] = col0 < G;  // G % 8 == 0 (checked on the host)
      if (valid[sw]) {
        if (Ly.perm == nullptr) {
          s_raw[sw] = *(const u32x4*)(Ly.scale + col0);
          b_raw[sw] = *(const u32x4*)(Ly.wbias + col0);
#pragma unroll
          for (int t = 0; t < TOK; ++t)
            if (t < tokens) xs_raw[sw][t] = *(const u32x4*)(Ly.x + (size_t)t * G + col0);
        } else {
          // column c of the quantised matrix multiplies input feature perm[c]
          const u32x4 pv = *(const u32x4*)(Ly.perm + col0);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const uint32_t j0 = pv[q] & 0xffffu, j1 = pv[q] >> 16;
            s_raw[sw][q] = (uint32_t)Ly.scale[j0] | ((uint32_t)Ly.scale[j1] << 16);
            b_raw[sw][q] = (uint32_t)Ly.wbias[j0] | ((uint32_t)Ly.wbias[j1] << 16);
#pragma unroll
            for (int t = 0; t < TOK; ++t)
              if (t < tokens)
                xs_raw[sw][t][q] = (uint32_t)Ly.x[(size_t)t * G + j0] |
                                   ((uint32_t)Ly.x[(size_t)t * G + j1] << 16);
          }
        }
      }
    }
#pragma unroll
    for (int sw = 0; sw < kSW; ++sw) {
      const int col0 = base + sw * kSweepCols + tid * 8;
#pragma unroll
      for (int r = 0; r < ROWS; ++r) {
        // rows past N (last workgroup) re-read row N-1; their results are dropped
        const int row = row0 + r < N ? row0 + r : N - 1;
        if (valid[sw])
          iw[sw][r] = *(const u32x4*)(Ly.idx + (size_t)row * Ly.row_words + (col0 >> 1));
      }
    }

    // ---- 3. build the LDS codebook image (first iteration only) ----
    if (!tables_ready) {
      tables_ready = true;
      if (TAB == 1) {
        // thread t owns the 256-byte bank row of entry t: 16 replicas.  The
        // replica order is rotated by the lane id so the 8 lanes of a
        // ds_write_b128 group hit 8 different 16-byte slots.
        const uint32_t rowp = (tid >> 8) * L::kTableBytes + (tid & 255) * 256;
#pragma unroll
        for (int q = 0; q < 16; ++q) lds_store16(rowp + (((q + lane) & 15) << 4), centry);
      } else {
        lds_store16((tid >> 8) * L::kTableBytes + (tid & 255) * 16, centry);
      }
      __syncthreads();
    }

    // ---- 4. dequantise + accumulate ----
#pragma unroll
    for (int sw = 0; sw < kSW; ++sw) {
      if (!valid[sw]) continue;
      // per-column scale / bias pairs and activations for this sweep
      uint32_t s2[8], b2[8];
      float xf[TOK][8];
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const uint32_t sp = s_raw[sw][c >> 1], bp = b_raw[sw][c >> 1];
        const uint16_t sv = (c & 1) ? (uint16_t)(sp >> 16) : (uint16_t)sp;
        const uint16_t bv = (c & 1) ? (uint16_t)(bp >> 16) : (uint16_t)bp;
        s2[c] = splat16(sv);
        b2[c] = splat16(bv);
#pragma unroll
        for (int t = 0; t < TOK; ++t) {
          float xv = 0.f;
          if (t < tokens) {
            const uint32_t xp = xs_raw[sw][t][c >> 1];
            xv = (c & 1) ? DT::hi(xp) : DT::lo(xp);
          }
          if (FAST) {
            // folded form: y = sum (x*s)*(c+r) + sum x*b   (fp32)
            accb[t] = __builtin_fmaf(xv, DT::to_float(bv), accb[t]);
            xv = xv * DT::to_float(sv);
          }
          xf[t][c] = xv;
        }
      }
      // ROWS*2 half-groups of 4 elements; the gathers of half-group g+1 are
      // issued before the arithmetic of half-group g (LDS latency ~ one
      // half-group of VALU work).
      constexpr int NH = ROWS * 2;
      u32x4 cv[2][4], rv[2][4];
#pragma unroll
      for (int e = 0; e < 4; ++e)
        gather<TAB>(iw[sw][0][e >> 1], e & 1, baseC, baseR, cv[0][e], rv[0][e]);
#pragma unroll
      for (int g = 0; g < NH; ++g) {
        const int r = g >> 1, kh = g & 1;
        if (g + 1 < NH) {
          const int r1 = (g + 1) >> 1, kh1 = (g + 1) & 1;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            gather<TAB>(iw[sw][r1][kh1 * 2 + (e >> 1)], e & 1, baseC, baseR, cv[(g + 1) & 1][e],
                        rv[(g + 1) & 1][e]);
        }
        // stage-major order: the 16 independent pair-chains of the half-group
        // advance together, so dependent packed-f16 ops are never back-to-back
        uint32_t w2[4][4];
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int p = 0; p < 4; ++p) w2[e][p] = DT::add2(cv[g & 1][e][p], rv[g & 1][e][p]);
        if (!FAST) {
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int p = 0; p < 4; ++p) w2[e][p] = DT::mul2(w2[e][p], s2[kh * 4 + e]);
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int p = 0; p < 4; ++p) w2[e][p] = DT::add2(w2[e][p], b2[kh * 4 + e]);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int c = kh * 4 + e;
#pragma unroll
          for (int t = 0; t < TOK; ++t)
#pragma unroll
            for (int p = 0; p < 4; ++p) {
              acc[t][r][2 * p] = DT::fma_lo(w2[e][p], xf[t][c], acc[t][r][2 * p]);
              acc[t][r][2 * p + 1] = DT::fma_hi(w2[e][p], xf[t][c], acc[t][r][2 * p + 1]);
            }
        }
      }
    }
  }

  // ---- 5. reduce over the workgroup's lanes and store ----
  constexpr int kVals = TOK * ROWS * 8;
  constexpr int kStride = kVals + TOK;  // floats per wave in the scratch area
  const uint32_t red0 = L::kScratchOff;
  float* red = (float*)(smem + L::kScratchOff);  // [kWaves][kStride]
  if (RED == 1) {
    float v[kVals];
#pragma unroll
    for (int t = 0; t < TOK; ++t)
#pragma unroll
      for (int r = 0; r < ROWS; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) v[(t * ROWS + r) * 8 + i] = acc[t][r][i];
    using WR = WaveReduce<kVals>;
    WR::run(v);
    const int q = lane >> 4;
    if ((lane & 15) == 0 && WR::writer(q)) {
      const uint32_t a = red0 + (wave * kStride + WR::base(q)) * 4;
#pragma unroll
      for (int i = 0; i < WR::kM; i += 4)
        lds_store16(a + i * 4, u32x4{__float_as_uint(v[i]), __float_as_uint(v[i + 1]),
                                     __float_as_uint(v[i + 2]), __float_as_uint(v[i + 3])});
    }
  } else {
#pragma unroll
    for (int t = 0; t < TOK; ++t)
#pragma unroll
      for (int r = 0; r < ROWS; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float sum = wave_sum(acc[t][r][i]);
          if (lane == 0) red[wave * kStride + (t * ROWS + r) * 8 + i] = sum;
        }
  }
  if (FAST) {
#pragma unroll
    for (int t = 0; t < TOK; ++t) {
      const float sum = wave_sum(accb[t]);
      if (lane == 0) red[wave * kStride + kVals + t] = sum;
    }
  }
  __syncthreads();
  if (tid < kVals) {
    const int t = tid / (ROWS * 8), rem = tid - t * (ROWS * 8);
    const int row = row0 + (rem >> 3);
    const int o = row * 8 + (rem & 7);
    if (t < tokens && row < N && o < O) {
      float sum = 0.f;
#pragma unroll
      for (int w = 0; w < kWaves; ++w) {
        sum += red[w * kStride + tid];
        if (FAST) sum += red[w * kStride + kVals + t];
      }
      if (Ly.bias) sum += DT::to_float(Ly.bias[o]);
      Ly.y[(size_t)t * O + o] = DT::from_float(sum);
    }
  }
}

// ---- host side -------------------------------------------------------------------
bool gemv_k256_eligible(const VptqLayerDesc& d, int tokens) {
  return d.vector_len == 8 && d.num_centroids == 256 && d.num_res_centroids == 256 &&
         d.index_bits == 8 && d.res_bits == 8 && d.num_codebooks == 1 && d.outlier_size == 0 &&
         d.weight_scale != nullptr && d.weight_bias != nullptr && (d.group_size % 8) == 0 &&
         d.group_size == d.in_features && d.row_words == d.group_size / 2 && tokens >= 1 &&
         tokens <= 4 && (((uintptr_t)d.indices | (uintptr_t)d.centroids |
                          (uintptr_t)d.res_centroids | (uintptr_t)d.weight_scale |
                          (uintptr_t)d.weight_bias | (uintptr_t)d.perm) & 15) == 0;
}

static int tab_mode() {
  static int m = -1;
  if (m < 0) {
    const char* e = getenv("VPTQ_K256_TAB");
    m = (e && e[0] == '0') ? 0 : 1;
  }
  return m;
}

static int red_mode() {  // VPTQ_K256_REDUCE=0: plain shuffle tree (A/B, debugging)
  static int m = -1;
  if (m < 0) {
    const char* e = getenv("VPTQ_K256_REDUCE");
    m = (e && e[0] == '0') ? 0 : 1;
  }
  return m;
}

// ROWS is chosen so the grid is about one workgroup per CU (256 CUs).
static int pick_rows(int n_rows_total, int tokens, bool f16) {
  int rows = 4;
  while (rows > 1 && (n_rows_total + rows - 1) / rows < 256) rows >>= 1;
  // accumulators per lane = 8 * rows * tok: keep the kernel spill-free
  const int tok = tokens > 2 ? 4 : tokens;
  const int cap = f16 ? (tok == 4 ? 4 : 8) : 4;
  while (rows > 1 && rows * tok > cap) rows >>= 1;
  return rows;
}

template <typename DT, int ROWS, int TOK, bool FAST, int TAB, int RED>
static hipError_t launch_inst(const K256Params& P, int grid, hipStream_t st) {
  auto kern = gemv_k256_kernel<DT, ROWS, TOK, FAST, TAB, RED>;
  constexpr int lds = Lds<TAB>::kScratchOff + kWaves * (TOK * ROWS * 8 + TOK) * 4;
  static bool attr_set = false;  // benign race: idempotent
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), lds, st, P);
  return hipGetLastError();
}

#define K256_CASE(DT, R, T, F, TB)                                   \
  if (rows == R && tok == T && fast == F && tab == TB) {               \
    if (red) return launch_inst<DT, R, T, F, TB, 1>(P, grid, st);      \
    return launch_inst<DT, R, T, F, TB, 0>(P, grid, st);               \
  }

template <typename DT, bool ALLOW_FAST>
static hipError_t dispatch(const K256Params& P, int grid, int rows, int tok, bool fast, int tab,
                           int red, hipStream_t st) {
  K256_CASE(DT, 1, 1, false, 1) K256_CASE(DT, 2, 1, false, 1) K256_CASE(DT, 4, 1, false, 1)
  K256_CASE(DT, 1, 2, false, 1) K256_CASE(DT, 2, 2, false, 1)
  K256_CASE(DT, 1, 4, false, 1)
  if constexpr (ALLOW_FAST) { K256_CASE(DT, 4, 2, false, 1) }
  if constexpr (ALLOW_FAST) {
    K256_CASE(DT, 1, 1, true, 1) K256_CASE(DT, 2, 1, true, 1) K256_CASE(DT, 4, 1, true, 1)
    K256_CASE(DT, 1, 2, true, 1) K256_CASE(DT, 2, 2, true, 1) K256_CASE(DT, 4, 2, true, 1)
    K256_CASE(DT, 1, 4, true, 1)
    // plain (un-replicated) tables: A/B baseline, tokens == 1 only
    K256_CASE(DT, 1, 1, false, 0) K256_CASE(DT, 2, 1, false, 0) K256_CASE(DT, 4, 1, false, 0)
    K256_CASE(DT, 1, 1, true, 0) K256_CASE(DT, 2, 1, true, 0) K256_CASE(DT, 4, 1, true, 0)
  }
  return hipErrorInvalidValue;
}

const char* gemv_k256_name(const VptqLayerDesc& d, int tokens, int flags) {
  (void)d; (void)tokens;
  return (flags & VPTQ_GEMV_FAST_MATH) ? "gemv_k256_kernel<fast>" : "gemv_k256_kernel";
}

hipError_t launch_gemv_k256(const VptqLayerDesc* descs, int n, const void* const* x,
                            void* const* y, int tokens, int flags, hipStream_t st) {
  if (n < 1 || n > kMaxGroup) return hipErrorInvalidValue;
  K256Params P;
  P.n_layers = n;
  P.tokens = tokens;
  int total_rows = 0;
  for (int i = 0; i < n; ++i) total_rows += descs[i].num_indices;
  const int tok = tokens > 2 ? 4 : tokens;
  const int rows = pick_rows(total_rows, tokens, descs[0].dtype == VPTQ_DTYPE_F16);
  int grid = 0;
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
    Ly.perm = d.perm;
    Ly.N = d.num_indices;
    Ly.G = d.group_size;
    Ly.O = d.out_features;
    Ly.row_words = d.row_words;
    Ly.wg_begin = grid;
    Ly.pad_ = 0;
    grid += (d.num_indices + rows - 1) / rows;
  }
  const bool f16 = descs[0].dtype == VPTQ_DTYPE_F16;
  const bool fast = f16 && (flags & VPTQ_GEMV_FAST_MATH);
  int tab = tab_mode();
  if (!f16 || tok != 1) tab = 1;
  return f16 ? dispatch<F16, true>(P, grid, rows, tok, fast, tab, red_mode(), st)
             : dispatch<BF16, false>(P, grid, rows, tok, false, tab, red_mode(), st);
}

}  // namespace vptq
