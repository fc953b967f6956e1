// VPTQ_GEMV_SELECTIVE over the sliced layouts of the large-codebook formats (round 6): the pre-pass of vptq_quant_gemv_sliced(...,
// flags | VPTQ_GEMV_SELECTIVE, ...).  The folded kernel over the layouts (gemv_sliced.hip) evaluates y = sum (c + r) f16(s x) + sum b x;
// its distance to the reference's roundings - w = f16(f16(f16(c + r) s) + b), vptq/ops/quant_gemm.py:121,155-156 - is a sum of
// per-column rounding errors that averages out unless a handful of activation columns dominate the token (gemv_k256c.hip has the
// story and the counts).  This launch, in front of the folded one:
//   * finds the threshold - kappa x rms of f16(s x) over the layer's columns - and the HOT blocks of 128 columns (one of their
//     columns at or above it): every workgroup for itself, 48 KiB out of the L2;
//   * workgroup 0 writes x_masked: the activation with the hot blocks' input features zeroed - what the folded launch reads, so
//     those columns add nothing there (neither to the gathers' products nor to sum b x);
//   * every workgroup rebuilds the hot blocks' weights of ITS rows as the reference rounds them - index words out of the packed
//     tensor (any element width up to 32 bits), entries gathered from the codebooks in device memory: few columns, the caches
//     hold what they touch - and stores sum w x per output (float32, zeros without a hot block) as `corr`, which the folded launch
//     adds before its one rounding (SlicedParams::corr).
// Replaces, for those columns, the reference's global gathers of both entries (csrc/kernels/quant_gemv.cuh:114-125).
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace vptq {

constexpr int kGHRows = 16;            // vector-rows per workgroup (16 lanes per row, 8 columns per lane and hot block): the exact products are
                                       // latency-bound gathers out of the L2 - as many workgroups as the rows give (64 rows per workgroup:
                                       // 16 workgroups on 256 CUs, 130 us per call inside a decoder with a handful of hot blocks per layer)
constexpr int kGHMaxHot = 16;          // hot blocks the corrections cover: with more, the threshold is raised until the 16 most dominant
                                       // remain (a token with dozens of "dominant" blocks is a dense one: the folded form's regime)
constexpr int kGHMaxBlocks = 256;      // 32768 columns

struct HotParams {
  const uint32_t* idx;
  const uint32_t* cent;
  const uint32_t* rcent;     // null: no residual codebook
  const uint16_t* x;
  const uint16_t* scale;     // column order
  const uint16_t* cbias;     // column order
  const uint16_t* perm;      // column c multiplies input feature perm[c], or null
  uint16_t* xm;              // [I] out: x with the hot blocks' features zeroed
  float* corr;               // [N x V] out
  uint32_t* hdr;             // {threshold bits, hot blocks}
  int N, G, O, row_words, T;
  float kappa;
};

template <typename DT, int V>
__global__ __launch_bounds__(256) void gemv_hot_kernel(const HotParams P) {
  __shared__ float part[4];
  __shared__ uint32_t mask[kGHMaxBlocks / 32];
  const int tid = (int)threadIdx.x;
  const uint16_t* const x = as_global(P.x);
  const uint16_t* const sc = as_global(P.scale);
  const uint16_t* const pm = as_global(P.perm);
  if (tid < kGHMaxBlocks / 32) mask[tid] = 0u;
  // f16(s x) of columns c, c + 1 (c even)
  auto staged = [&](int c) -> uint32_t {
    uint32_t xv;
    if (pm) {
      const uint32_t pv = *(const uint32_t*)(pm + c);
      xv = (uint32_t)x[pv & 0xffffu] | ((uint32_t)x[pv >> 16] << 16);
    } else {
      xv = *(const uint32_t*)(x + c);
    }
    return DT::mul2(xv, *(const uint32_t*)(sc + c));
  };
  constexpr int kPer = 16;   // pairs per thread and batch (8192 columns per batch): requested together; the first batch stays in
  uint32_t xp0[kPer];        // registers for the second pass
  float ss = 0.f;
  for (int base = 0; base < P.G; base += 512 * kPer) {
    uint32_t xp[kPer];
#pragma unroll
    for (int i = 0; i < kPer; ++i) {
      const int c = base + i * 512 + 2 * tid;
      xp[i] = c < P.G ? staged(c) : 0u;
    }
#pragma unroll
    for (int i = 0; i < kPer; ++i) {
      if (base == 0) xp0[i] = xp[i];
      const float a = DT::lo(xp[i]), b = DT::hi(xp[i]);
      ss = __builtin_fmaf(a, a, __builtin_fmaf(b, b, ss));
    }
  }
  ss = wave_sum(ss);
  if ((tid & 63) == 0) part[tid >> 6] = ss;
  __syncthreads();
  uint32_t thr = DT::kInfBits;                     // inf / NaN sums: only inf / NaN columns are hot
  {
    const float tot = (part[0] + part[1]) + (part[2] + part[3]);
    const float t = P.kappa * __builtin_sqrtf(tot / (float)P.G);
    if (t < DT::kMaxFinite) {
      thr = (uint32_t)DT::from_float(t) & 0x7fffu;
      if (DT::to_float((uint16_t)thr) < t) thr += 1u;
    }
    if (thr == 0u) thr = 1u;                       // (all-zero activations: nothing is hot)
  }
  for (int base = 0; base < P.G; base += 512 * kPer) {
    uint32_t xp[kPer];
#pragma unroll
    for (int i = 0; i < kPer; ++i) {
      const int c = base + i * 512 + 2 * tid;
      xp[i] = base == 0 ? xp0[i] : (c < P.G ? staged(c) : 0u);
    }
#pragma unroll
    for (int i = 0; i < kPer; ++i) {
      const int c = base + i * 512 + 2 * tid;
      if ((xp[i] & 0x7fffu) >= thr || ((xp[i] >> 16) & 0x7fffu) >= thr) atomicOr(&mask[c >> 12], 1u << ((c >> 7) & 31));
    }
  }
  __syncthreads();
  int nhot = 0;
#pragma unroll
  for (int i = 0; i < kGHMaxBlocks / 32; ++i) nhot += __builtin_popcount(mask[i]);
  // too many: raise the threshold by a quarter (magnitude bits of a 16-bit float: + 1/4 of an octave per 256 / 32 ... simply scale
  // the float) and look again - every workgroup walks the same sequence
  for (int it = 0; it < 24 && nhot > kGHMaxHot; ++it) {
    __syncthreads();
    if (tid < kGHMaxBlocks / 32) mask[tid] = 0u;
    __syncthreads();
    const float tf = DT::to_float((uint16_t)thr) * 1.25f;
    thr = tf < DT::kMaxFinite ? ((uint32_t)DT::from_float(tf) & 0x7fffu) + 1u : DT::kInfBits;
    for (int base = 0; base < P.G; base += 512 * kPer) {
#pragma unroll
      for (int i = 0; i < kPer; ++i) {
        const int c = base + i * 512 + 2 * tid;
        const uint32_t xq = base == 0 ? xp0[i] : (c < P.G ? staged(c) : 0u);
        if ((xq & 0x7fffu) >= thr || ((xq >> 16) & 0x7fffu) >= thr) atomicOr(&mask[c >> 12], 1u << ((c >> 7) & 31));
      }
    }
    __syncthreads();
    nhot = 0;
#pragma unroll
    for (int i = 0; i < kGHMaxBlocks / 32; ++i) nhot += __builtin_popcount(mask[i]);
  }
  if (blockIdx.x == 0 && tid == 0) { as_global(P.hdr)[0] = thr; as_global(P.hdr)[1] = (uint32_t)nhot; }
  {
    // x_masked, shared out over ALL workgroups in chunks of 8 columns (one workgroup writing it alone was 32 dependent round trips:
    // 18 us); a permutation is a bijection - every feature is written exactly once
    uint16_t* const xm = as_global(P.xm);
    for (int ch = (int)blockIdx.x * 256 + tid; ch * 8 < P.G; ch += (int)gridDim.x * 256) {
      const int c = ch * 8;
      const bool hot = ((mask[c >> 12] >> ((c >> 7) & 31)) & 1u) != 0u;
      if (pm) {
        const u32x4 pv = *(const u32x4*)(pm + c);
        uint16_t v[8];
#pragma unroll
        for (int q = 0; q < 4; ++q) { v[2 * q] = x[pv[q] & 0xffffu]; v[2 * q + 1] = x[pv[q] >> 16]; }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          xm[pv[q] & 0xffffu] = hot ? (uint16_t)0 : v[2 * q];
          xm[pv[q] >> 16] = hot ? (uint16_t)0 : v[2 * q + 1];
        }
      } else {
        const u32x4 v = *(const u32x4*)(x + c);
        *(u32x4*)(xm + c) = hot ? u32x4{0u, 0u, 0u, 0u} : v;
      }
    }
  }
  if ((int)blockIdx.x * kGHRows >= P.N) return;
  const int sub = tid & 15;                        // 8 columns of the block
  const uint32_t tmask = P.T >= 32 ? 0xffffffffu : ((1u << P.T) - 1u);
  const uint16_t* const cb = as_global(P.cbias);
  const u32x4* const cent = (const u32x4*)as_global(P.cent);
  const u32x4* const rcent = (const u32x4*)as_global(P.rcent);
#pragma unroll 1
  for (int round = 0; round < kGHRows / 16; ++round) {
    const int row = (int)blockIdx.x * kGHRows + round * 16 + (tid >> 4);
    float acc[V];
#pragma unroll
    for (int k = 0; k < V; ++k) acc[k] = 0.f;
    const int rr = row < P.N ? row : P.N - 1;
    const uint32_t* const irow = as_global(P.idx) + (size_t)rr * (size_t)P.row_words;
    if (nhot != 0) {
      for (int wd = 0; wd < kGHMaxBlocks / 32; ++wd) {
        uint32_t m = mask[wd];
        while (m) {
          const int bit = __builtin_ctz(m);
          m &= m - 1u;
          const int c0 = (wd * 32 + bit) * 128 + sub * 8;
          if (c0 >= P.G) continue;                 // (G is a multiple of 8)
#pragma unroll
          for (int j = 0; j < 8; j += 2) {
            const int c = c0 + j;
            const uint32_t sp = *(const uint32_t*)(sc + c), bp = *(const uint32_t*)(cb + c);
            uint32_t xw;
            if (pm) {
              const uint32_t pv = *(const uint32_t*)(pm + c);
              xw = (uint32_t)x[pv & 0xffffu] | ((uint32_t)x[pv >> 16] << 16);
            } else {
              xw = *(const uint32_t*)(x + c);
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const long long bitpos = (long long)(c + h) * P.T;
              const int wi = (int)(bitpos >> 5), sh = (int)(bitpos & 31);
              const uint32_t lo = irow[wi], hi = irow[wi + 1 < P.row_words ? wi + 1 : wi];
              const uint32_t e = (uint32_t)((((unsigned long long)hi << 32) | lo) >> sh) & tmask;
              const uint32_t ci = e & 0xffffu, ri = e >> 16;
              const float xf = DT::half_of(xw, h);
#pragma unroll
              for (int q = 0; q < V / 8; ++q) {
                const u32x4 ce = cent[(size_t)ci * (V / 8) + q];
                u32x4 re = u32x4{0u, 0u, 0u, 0u};
                if (rcent) re = rcent[(size_t)ri * (V / 8) + q];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                  uint32_t w = rcent ? DT::add2(ce[k], re[k]) : ce[k];
                  w = DT::mul2_bcast(w, sp, h);
                  w = DT::add2_bcast(w, bp, h);
                  acc[8 * q + 2 * k] = DT::fma_lo(w, xf, acc[8 * q + 2 * k]);
                  acc[8 * q + 2 * k + 1] = DT::fma_hi(w, xf, acc[8 * q + 2 * k + 1]);
                }
              }
            }
          }
        }
      }
#pragma unroll
      for (int k = 0; k < V; ++k) acc[k] = row16_allsum(acc[k]);
    }
    if (sub == 0 && row < P.N) {
      float* const co = as_global(P.corr) + (size_t)row * V;
#pragma unroll
      for (int k = 0; k < V; ++k)
        if (row * V + k < P.O) co[k] = acc[k];
    }
  }
}

// ---- host side -------------------------------------------------------------------
// layers the pre-pass serves: what the folded sliced kernel serves, with scale and bias, element words of <= 32 bits
bool gemv_hot_eligible(const VptqLayerDesc& d) {
  return gemv_sliced_eligible(d, false) && d.weight_scale && d.weight_bias &&
         d.num_codebooks == 1 && d.outlier_size == 0 && d.index_bits == 16 && d.index_bits + d.res_bits <= 32 &&
         (d.vector_len == 8 || d.vector_len == 16) && d.group_size <= kGHMaxBlocks * 128 && (d.group_size % 8) == 0 &&
         (d.perm == nullptr || (d.scale_permuted && d.bias_permuted));
}
// bytes behind the accumulator words of vptq_quant_gemv_sliced's workspace: header | x_masked | corr
size_t gemv_hot_bytes(const VptqLayerDesc& d) {
  return 256 + ((size_t)d.in_features * 2 + 255) / 256 * 256 + ((size_t)d.num_indices * d.vector_len * 4 + 255) / 256 * 256;
}
static float gh_kappa() {
  static std::atomic<int> milli{-1};
  if (milli < 0) { const char* e = tune_env("VPTQ_SELECTIVE_KAPPA"); const double v = e ? atof(e) : 0.0; milli = v > 0.0 ? (int)(v * 1000.0) : 6000; }
  return (float)milli.load() * 1e-3f;
}
// extra = gemv_hot_bytes(d) bytes, 256-byte aligned; returns where x_masked and corr are
hipError_t launch_gemv_hot(const VptqLayerDesc& d, const void* x, void* extra, const void** x_masked, const float** corr, hipStream_t st) {
  if (!gemv_hot_eligible(d) || !extra || (((uintptr_t)extra) & 255) != 0 || (((uintptr_t)x) & 15) != 0) return hipErrorInvalidValue;
  HotParams P;
  P.idx = (const uint32_t*)d.indices;
  P.cent = (const uint32_t*)d.centroids;
  P.rcent = d.num_res_centroids > 0 ? (const uint32_t*)d.res_centroids : nullptr;
  P.x = (const uint16_t*)x;
  P.scale = (const uint16_t*)(d.perm ? d.scale_permuted : d.weight_scale);
  P.cbias = (const uint16_t*)(d.perm ? d.bias_permuted : d.weight_bias);
  P.perm = (const uint16_t*)d.perm;
  P.hdr = (uint32_t*)extra;
  P.xm = (uint16_t*)((char*)extra + 256);
  P.corr = (float*)((char*)extra + 256 + ((size_t)d.in_features * 2 + 255) / 256 * 256);
  P.N = d.num_indices; P.G = d.group_size; P.O = d.out_features; P.row_words = d.row_words;
  P.T = d.index_bits + d.res_bits;
  P.kappa = gh_kappa();
  if ((((uintptr_t)P.cent | (uintptr_t)P.rcent) & 15) != 0) return hipErrorInvalidValue;
  const int grid = (d.num_indices + kGHRows - 1) / kGHRows;
  const bool f16 = d.dtype == VPTQ_DTYPE_F16;
  if (d.vector_len == 8) {
    if (f16) hipLaunchKernelGGL((gemv_hot_kernel<F16, 8>), dim3(grid), dim3(256), 0, st, P);
    else hipLaunchKernelGGL((gemv_hot_kernel<BF16, 8>), dim3(grid), dim3(256), 0, st, P);
  } else {
    if (f16) hipLaunchKernelGGL((gemv_hot_kernel<F16, 16>), dim3(grid), dim3(256), 0, st, P);
    else hipLaunchKernelGGL((gemv_hot_kernel<BF16, 16>), dim3(grid), dim3(256), 0, st, P);
  }
  *x_masked = P.xm;
  *corr = P.corr;
  return hipGetLastError();
}

}  // namespace vptq
