// Dequantise a packed VQuantLinear layer to the dense W[O, I] the reference CPU
// path produces (vptq/ops/quant_gemm.py:43-158), bit for bit.
//
// Replaces DequantizeWithOutliers_PackIndice (reference csrc/kernels/dequant.cuh:9-115,
// launcher csrc/dequant.cu:154-225).  One thread per (vector-row n, output
// column j): lanes run along j so each of the v stores of a wave is one
// contiguous 128-byte segment of a W row; with a permutation the READ side is
// the gather (inv_perm[j] -> index element), the write side stays coalesced.
#include "common.h"
#include "kernels.h"

namespace vptq {

template <typename DT, int V>
__global__ __launch_bounds__(256) void dequant_kernel(const VptqLayerDesc d,
                                                      uint16_t* __restrict__ W) {
  constexpr int VP = V / 2;
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int n = blockIdx.y;
  const int I = d.in_features, O = d.out_features, S = d.outlier_size, G = d.group_size;
  if (j >= I) return;
  const int c = d.inv_perm ? (int)d.inv_perm[j] : j;
  const int T = d.index_bits + d.res_bits;
  uint32_t w2[VP];
  if (c < S) {
    const int ov = d.outlier_vector_len;
    const uint16_t* ocent = (const uint16_t*)d.outlier_centroids;
#pragma unroll
    for (int p = 0; p < VP; ++p) {
      uint32_t pr = 0;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int o = n * V + 2 * p + h;
        uint16_t e = 0;
        if (o < O) {
          const int m = o / ov, tt = o - m * ov;
          e = ocent[(size_t)d.outlier_indices[(size_t)m * S + c] * ov + tt];
        }
        pr |= (uint32_t)e << (16 * h);
      }
      w2[p] = pr;
    }
  } else {
    const int cc = c - S;
    const int cb = cc / G, g = cc - cb * G;
    const uint32_t* row =
        (const uint32_t*)d.indices + ((size_t)cb * d.num_indices + n) * d.row_words;
    const uint32_t e = unpack_elem(row, g, T);
    const uint32_t idx = e & ((1u << d.index_bits) - 1u);
    const uint32_t* cp = (const uint32_t*)d.centroids + ((size_t)cb * d.num_centroids + idx) * VP;
#pragma unroll
    for (int p = 0; p < VP; ++p) w2[p] = cp[p];
    if (d.res_bits) {
      const uint32_t ridx = (e >> d.index_bits) & ((1u << d.res_bits) - 1u);
      const uint32_t* rp =
          (const uint32_t*)d.res_centroids + ((size_t)cb * d.num_res_centroids + ridx) * VP;
#pragma unroll
      for (int p = 0; p < VP; ++p) w2[p] = DT::add2(w2[p], rp[p]);
    }
  }
  if (d.weight_scale) {
    const uint32_t s2 = splat16(((const uint16_t*)d.weight_scale)[j]);
    const uint32_t b2 = splat16(((const uint16_t*)d.weight_bias)[j]);
#pragma unroll
    for (int p = 0; p < VP; ++p) w2[p] = DT::add2(DT::mul2(w2[p], s2), b2);
  }
#pragma unroll
  for (int p = 0; p < VP; ++p) {
    const int o = n * V + 2 * p;
    if (o < O) W[(size_t)o * I + j] = (uint16_t)w2[p];
    if (o + 1 < O) W[(size_t)(o + 1) * I + j] = (uint16_t)(w2[p] >> 16);
  }
}

template <typename DT, int V>
static hipError_t launch_v(const VptqLayerDesc& d, void* W, hipStream_t st) {
  dim3 grid((d.in_features + 255) / 256, d.num_indices), block(256);
  hipLaunchKernelGGL((dequant_kernel<DT, V>), grid, block, 0, st, d, (uint16_t*)W);
  return hipGetLastError();
}

template <typename DT>
static hipError_t launch_dt(const VptqLayerDesc& d, void* W, hipStream_t st) {
  switch (d.vector_len) {
    case 2: return launch_v<DT, 2>(d, W, st);
    case 4: return launch_v<DT, 4>(d, W, st);
    case 6: return launch_v<DT, 6>(d, W, st);
    case 8: return launch_v<DT, 8>(d, W, st);
    case 10: return launch_v<DT, 10>(d, W, st);
    case 12: return launch_v<DT, 12>(d, W, st);
    case 16: return launch_v<DT, 16>(d, W, st);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_dequant(const VptqLayerDesc& d, void* W, hipStream_t st) {
  return d.dtype == VPTQ_DTYPE_F16 ? launch_dt<F16>(d, W, st) : launch_dt<BF16>(d, W, st);
}

}  // namespace vptq
