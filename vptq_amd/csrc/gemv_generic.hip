// Generic fused dequant+GEMV for every VQuantLinear configuration
// (any v in {2..16 even}, any index/residual bit width, C >= 1 codebooks,
// outlier columns, perm, optional norm/bias, f16/bf16, 1..8 tokens).
//
// Replaces WqA16WithOutliers_PackIndice (reference csrc/kernels/quant_gemv.cuh:11-186)
// for the configurations the specialised kernel (gemv_k256.hip) does not take.
// One 256-thread workgroup owns one vector-row n (v outputs) over ALL input
// columns, so there is no split-K partial buffer and no second reduction
// kernel (the reference needs tmp[..,O,I/1024] + tensor.sum, quant_gemv.cu:203-235).
// Codebooks are gathered through L1/L2; accumulation is fp32 (the reference
// accumulates in fp16, quant_gemv.cuh:137-142, which would fail the parity bar).
#include "common.h"
#include "kernels.h"

namespace vptq {

template <typename DT, int V, int TOK>
__global__ __launch_bounds__(256) void gemv_generic_kernel(const VptqLayerDesc d,
                                                           const uint16_t* __restrict__ x,
                                                           uint16_t* __restrict__ y, int tokens,
                                                           const int out_f32) {
  constexpr int VP = V / 2;
  const int n = blockIdx.x;
  const int tid = threadIdx.x;
  const int I = d.in_features, O = d.out_features, S = d.outlier_size, G = d.group_size;
  const int T = d.index_bits + d.res_bits;
  const uint32_t imask = (1u << d.index_bits) - 1u;
  const uint32_t rmask = d.res_bits ? ((1u << d.res_bits) - 1u) : 0u;
  const bool norm = d.weight_scale != nullptr;
  const uint16_t* scale = (const uint16_t*)d.weight_scale;
  const uint16_t* wbias = (const uint16_t*)d.weight_bias;
  const uint32_t* cent = (const uint32_t*)d.centroids;
  const uint32_t* rcent = (const uint32_t*)d.res_centroids;
  const uint16_t* ocent = (const uint16_t*)d.outlier_centroids;

  float acc[TOK][V];
#pragma unroll
  for (int t = 0; t < TOK; ++t)
#pragma unroll
    for (int i = 0; i < V; ++i) acc[t][i] = 0.f;

  // accumulate one rebuilt weight vector against the activations of input feature j
  auto accumulate = [&](const uint32_t (&w2)[VP], int j) {
#pragma unroll
    for (int t = 0; t < TOK; ++t) {
      if (t < tokens) {
        const float xf = DT::to_float(x[(size_t)t * I + j]);
#pragma unroll
        for (int p = 0; p < VP; ++p) {
          acc[t][2 * p] = DT::fma_lo(w2[p], xf, acc[t][2 * p]);
          acc[t][2 * p + 1] = DT::fma_hi(w2[p], xf, acc[t][2 * p + 1]);
        }
      }
    }
  };
  auto normalise = [&](uint32_t (&w2)[VP], int j) {
    if (norm) {
      const uint32_t s2 = splat16(scale[j]), b2 = splat16(wbias[j]);
#pragma unroll
      for (int p = 0; p < VP; ++p) w2[p] = DT::add2(DT::mul2(w2[p], s2), b2);
    }
  };

  // ---- outlier columns (c < S): W[m*ov+tt, c] = outlier_centroids[oidx[m, c], tt] ----
  for (int c = tid; c < S; c += 256) {
    const int j = d.perm ? (int)d.perm[c] : c;
    const int ov = d.outlier_vector_len;
    uint32_t w2[VP];
#pragma unroll
    for (int p = 0; p < VP; ++p) {
      uint32_t pr = 0;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int o = n * V + 2 * p + h;
        uint16_t e = 0;
        if (o < O) {
          const int m = o / ov, tt = o - m * ov;
          const uint32_t oi = d.outlier_indices[(size_t)m * S + c];
          e = ocent[(size_t)oi * ov + tt];
        }
        pr |= (uint32_t)e << (16 * h);
      }
      w2[p] = pr;
    }
    normalise(w2, j);
    accumulate(w2, j);
  }

  // ---- codebook columns: kU independent index -> gather chains in flight per thread ----
  constexpr int kU = 4;
  const int IC = I - S;  // = C * G
  for (int c0 = tid; c0 < IC; c0 += 256 * kU) {
    uint32_t e[kU];
    int j[kU], cbv[kU];
    bool ok[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int cc = c0 + u * 256;
      ok[u] = cc < IC;
      const int ccl = ok[u] ? cc : IC - 1;          // clamped: loads stay in bounds
      const int cb = ccl / G, g = ccl - cb * G;
      cbv[u] = cb;
      const uint32_t* row =
          (const uint32_t*)d.indices + ((size_t)cb * d.num_indices + n) * d.row_words;
      e[u] = unpack_elem(row, g, T);
      j[u] = d.perm ? (int)d.perm[S + ccl] : S + ccl;
    }
    uint32_t w2[kU][VP];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const uint32_t* cp = cent + ((size_t)cbv[u] * d.num_centroids + (e[u] & imask)) * VP;
#pragma unroll
      for (int p = 0; p < VP; ++p) w2[u][p] = cp[p];
    }
    if (rmask) {
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const uint32_t ridx = (e[u] >> d.index_bits) & rmask;
        const uint32_t* rp = rcent + ((size_t)cbv[u] * d.num_res_centroids + ridx) * VP;
#pragma unroll
        for (int p = 0; p < VP; ++p) w2[u][p] = DT::add2(w2[u][p], rp[p]);
      }
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      if (ok[u]) {
        normalise(w2[u], j[u]);
        accumulate(w2[u], j[u]);
      }
    }
  }

  __shared__ float red[4][TOK * V];
  const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
  for (int t = 0; t < TOK; ++t)
#pragma unroll
    for (int i = 0; i < V; ++i) {
      const float s = wave_sum(acc[t][i]);
      if (lane == 0) red[wave][t * V + i] = s;
    }
  __syncthreads();
  if (tid < TOK * V) {
    const int t = tid / V, i = tid - t * V;
    const int o = n * V + i;
    if (t < tokens && o < O) {
      float s = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
      if (d.bias) s += DT::to_float(((const uint16_t*)d.bias)[o]);
      if (out_f32) ((float*)y)[(size_t)t * O + o] = s;
      else y[(size_t)t * O + o] = DT::from_float(s);
    }
  }
}

template <typename DT, int V>
static hipError_t launch_v(const VptqLayerDesc& d, const void* x, void* y, int tokens, bool out_f32,
                           hipStream_t st) {
  dim3 grid(d.num_indices), block(256);
  const uint16_t* xp = (const uint16_t*)x;
  uint16_t* yp = (uint16_t*)y;
  if (tokens == 1)
    hipLaunchKernelGGL((gemv_generic_kernel<DT, V, 1>), grid, block, 0, st, d, xp, yp, tokens, (int)out_f32);
  else if (tokens == 2)
    hipLaunchKernelGGL((gemv_generic_kernel<DT, V, 2>), grid, block, 0, st, d, xp, yp, tokens, (int)out_f32);
  else if (tokens <= 4)
    hipLaunchKernelGGL((gemv_generic_kernel<DT, V, 4>), grid, block, 0, st, d, xp, yp, tokens, (int)out_f32);
  else
    hipLaunchKernelGGL((gemv_generic_kernel<DT, V, 8>), grid, block, 0, st, d, xp, yp, tokens, (int)out_f32);
  return hipGetLastError();
}

template <typename DT>
static hipError_t launch_dt(const VptqLayerDesc& d, const void* x, void* y, int tokens, bool out_f32,
                            hipStream_t st) {
  switch (d.vector_len) {
    case 2: return launch_v<DT, 2>(d, x, y, tokens, out_f32, st);
    case 4: return launch_v<DT, 4>(d, x, y, tokens, out_f32, st);
    case 6: return launch_v<DT, 6>(d, x, y, tokens, out_f32, st);
    case 8: return launch_v<DT, 8>(d, x, y, tokens, out_f32, st);
    case 10: return launch_v<DT, 10>(d, x, y, tokens, out_f32, st);
    case 12: return launch_v<DT, 12>(d, x, y, tokens, out_f32, st);
    case 16: return launch_v<DT, 16>(d, x, y, tokens, out_f32, st);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_gemv_generic(const VptqLayerDesc& d, const void* x, void* y, int tokens, bool out_f32,
                               hipStream_t st) {
  return d.dtype == VPTQ_DTYPE_F16 ? launch_dt<F16>(d, x, y, tokens, out_f32, st)
                                   : launch_dt<BF16>(d, x, y, tokens, out_f32, st);
}

}  // namespace vptq
