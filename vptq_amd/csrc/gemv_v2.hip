// quant_gemv_v2: fused dequant+GEMV on the UNPACKED v2 wire format
// (uint16 main ids laid out [N][I], uint8/uint16 residual ids, one codebook).
//
// Replaces ke_quant_gemv_v2 (reference csrc/kernels/quant_gemv_v2.cuh:15-184,
// host csrc/quant_gemv_v2.cu:25-180).  Format and expected result:
// reference tests/test_quant_gemv.py:49-109 (ground_truth).
// The reference restages the whole 128 KiB codebook into shared memory in every
// one of the N blocks; here codebook gathers go through L1/L2 (the k = 8192
// table cannot be LDS-replicated) and one workgroup owns a whole vector-row.
#include "common.h"
#include "kernels.h"

namespace vptq {

template <typename DT, int V, int TOK>
__global__ __launch_bounds__(256) void gemv_v2_kernel(const VptqV2Desc d,
                                                      const uint16_t* __restrict__ x,
                                                      uint16_t* __restrict__ y, int tokens,
                                                           const int out_f32) {
  constexpr int VP = V / 2;
  const int n = blockIdx.x, tid = threadIdx.x;
  const int I = d.in_features, O = d.out_features;
  const uint16_t* ids = d.indices + (size_t)n * I;
  const uint32_t* cent = (const uint32_t*)d.centroids;
  const uint32_t* rcent = (const uint32_t*)d.res_centroids;
  const uint16_t* scale = (const uint16_t*)d.scale_weights;
  const uint16_t* sbias = (const uint16_t*)d.scale_bias;

  float acc[TOK][V];
#pragma unroll
  for (int t = 0; t < TOK; ++t)
#pragma unroll
    for (int i = 0; i < V; ++i) acc[t][i] = 0.f;

  for (int i = tid; i < I; i += 256) {
    uint32_t w2[VP];
    const uint32_t* cp = cent + (size_t)ids[i] * VP;
#pragma unroll
    for (int p = 0; p < VP; ++p) w2[p] = cp[p];
    if (d.num_res_centroids > 0) {
      const uint32_t ridx = d.res_index_bytes == 1
                                ? (uint32_t)((const uint8_t*)d.res_indices)[(size_t)n * I + i]
                                : (uint32_t)((const uint16_t*)d.res_indices)[(size_t)n * I + i];
      const uint32_t* rp = rcent + (size_t)ridx * VP;
#pragma unroll
      for (int p = 0; p < VP; ++p) w2[p] = DT::add2(w2[p], rp[p]);
    }
    if (scale) {
      const uint32_t s2 = splat16(scale[i]);
#pragma unroll
      for (int p = 0; p < VP; ++p) w2[p] = DT::mul2(w2[p], s2);
    }
    if (sbias) {
      const uint32_t b2 = splat16(sbias[i]);
#pragma unroll
      for (int p = 0; p < VP; ++p) w2[p] = DT::add2(w2[p], b2);
    }
#pragma unroll
    for (int t = 0; t < TOK; ++t) {
      if (t < tokens) {
        const float xf = DT::to_float(x[(size_t)t * I + i]);
#pragma unroll
        for (int p = 0; p < VP; ++p) {
          acc[t][2 * p] = DT::fma_lo(w2[p], xf, acc[t][2 * p]);
          acc[t][2 * p + 1] = DT::fma_hi(w2[p], xf, acc[t][2 * p + 1]);
        }
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
static hipError_t launch_v(const VptqV2Desc& d, const void* x, void* y, int tokens, bool out_f32,
                           hipStream_t st) {
  dim3 grid((d.out_features + V - 1) / V), block(256);
  const uint16_t* xp = (const uint16_t*)x;
  uint16_t* yp = (uint16_t*)y;
  if (tokens == 1)
    hipLaunchKernelGGL((gemv_v2_kernel<DT, V, 1>), grid, block, 0, st, d, xp, yp, tokens, (int)out_f32);
  else if (tokens == 2)
    hipLaunchKernelGGL((gemv_v2_kernel<DT, V, 2>), grid, block, 0, st, d, xp, yp, tokens, (int)out_f32);
  else if (tokens <= 4)
    hipLaunchKernelGGL((gemv_v2_kernel<DT, V, 4>), grid, block, 0, st, d, xp, yp, tokens, (int)out_f32);
  else
    hipLaunchKernelGGL((gemv_v2_kernel<DT, V, 8>), grid, block, 0, st, d, xp, yp, tokens, (int)out_f32);
  return hipGetLastError();
}

template <typename DT>
static hipError_t launch_dt(const VptqV2Desc& d, const void* x, void* y, int tokens, bool out_f32,
                            hipStream_t st) {
  switch (d.vector_len) {
    case 4: return launch_v<DT, 4>(d, x, y, tokens, out_f32, st);
    case 8: return launch_v<DT, 8>(d, x, y, tokens, out_f32, st);
    case 16: return launch_v<DT, 16>(d, x, y, tokens, out_f32, st);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_gemv_v2(const VptqV2Desc& d, const void* x, void* y, int tokens, bool out_f32,
                          hipStream_t st) {
  return d.dtype == VPTQ_DTYPE_F16 ? launch_dt<F16>(d, x, y, tokens, out_f32, st)
                                   : launch_dt<BF16>(d, x, y, tokens, out_f32, st);
}

}  // namespace vptq
