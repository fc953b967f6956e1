// Dequantise a packed VQuantLinear layer to the dense W[O, I] the reference CPU
// path produces (vptq/ops/quant_gemm.py:43-158), bit for bit.
//
// Replaces DequantizeWithOutliers_PackIndice (reference csrc/kernels/dequant.cuh:9-115,
// launcher csrc/dequant.cu:154-225).  Lanes run along the output columns; with a
// permutation the READ side is the gather (inv_perm[j] -> index element), the write side
// stays coalesced.
#include "common.h"
#include "kernels.h"

namespace vptq {

// One thread = 8 consecutive output columns j0..j0+7 of one vector-row n: 8 index elements
// (contiguous in the bit stream unless a permutation scatters them), 8 centroid gathers, and
// after an in-register transposition (one v_perm_b32 per output dword) V stores of 16 bytes -
// a wave writes 1 KiB contiguous per W row.  (The first version stored 2 bytes per lane: 128 B
// per wave-instruction, 1.9-2.8 TB/s.)  Rows run along blockIdx.x together with the column
// blocks, so neither dimension meets the 65535 limit of grid.y / grid.z.
// LDSTAB: one codebook group whose two tables together are <= 16 KiB (the canonical 2-bit format:
// 8 KiB) - every workgroup copies them into LDS first and gathers from there.  The 16 random
// 16-byte gathers per thread through L1 set the pace of the first version (one lane address per clock
// in the vector-memory pipe: 55 us per 8192^2 layer, 2.4 TB/s of the 6 TB/s the stores could take).
// TAB = 2: only the residual table (<= 16 KiB, e.g. the 4 KiB of "v8-k65536-256") goes to LDS, the main
// table is gathered through L1 / L2.
// Index elements: 16-bit elements come as one 16-byte piece (vec_idx); any other width as two windows of
// 4 elements = 4 T bits each, fetched with one 16-byte load at 4-byte alignment (+ a fifth word where the
// width needs it), shifted to bit 0 once, elements at wave-uniform positions (the scheme of
// gemv_gatherx.hip) - instead of two dword loads per element: for "v8-k65536-256" the kernel issued 42
// lane addresses per 8 elements (16 of them index words, 8 residual gathers) and ran at exactly that pace,
// 84 us per 8192^2 layer.
constexpr int kDqLdsMax = 16384;

typedef uint32_t dq_u32_a4 __attribute__((aligned(4)));
typedef uint32_t dq_u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));

template <typename DT, int V, int TAB>
__global__ __launch_bounds__(256) void dequant_kernel(const VptqLayerDesc d,
                                                      uint16_t* __restrict__ W, const int col_blocks) {
  constexpr bool LDSTAB = TAB == 1;
  constexpr bool RESLDS = TAB == 2;
  constexpr int VP = V / 2;
  const int n = blockIdx.x / col_blocks;
  const int j0 = ((blockIdx.x - n * col_blocks) * 256 + threadIdx.x) * 8;
  const int I = d.in_features, O = d.out_features, S = d.outlier_size, G = d.group_size;
  extern __shared__ __attribute__((aligned(16))) uint32_t dq_tab[];   // [k][VP] | [kr][VP]
  if constexpr (LDSTAB) {
    const int nm = d.num_centroids * VP, nr = d.num_res_centroids * VP;
    const uint32_t* cm = (const uint32_t*)d.centroids;
    const uint32_t* cr = (const uint32_t*)d.res_centroids;
    for (int i = threadIdx.x; i < nm + nr; i += 256) dq_tab[i] = i < nm ? cm[i] : cr[i - nm];
    __syncthreads();
  }
  if constexpr (RESLDS) {
    const int nr = d.num_res_centroids * VP;
    const uint32_t* cr = (const uint32_t*)d.res_centroids;
    for (int i = threadIdx.x; i < nr; i += 256) dq_tab[i] = cr[i];
    __syncthreads();
  }
  if (j0 >= I) return;
  const int T = d.index_bits + d.res_bits;
  const bool full = j0 + 8 <= I && (I & 7) == 0;  // 16-byte aligned, whole chunk inside the row
  // scale / bias of the 8 output columns: contiguous whatever the permutation - two 16-byte loads
  // instead of sixteen 2-byte ones (the kernel is paced by lane addresses, not bytes)
  u32x4 sv8 = {0, 0, 0, 0}, bv8 = {0, 0, 0, 0};
  const bool vec_norm = full && d.weight_scale != nullptr &&
                        ((((uintptr_t)d.weight_scale | (uintptr_t)d.weight_bias) & 15) == 0);
  if (vec_norm) {
    sv8 = *(const u32x4*)((const uint16_t*)d.weight_scale + j0);
    bv8 = *(const u32x4*)((const uint16_t*)d.weight_bias + j0);
  }
  // 16-bit elements, no permutation, no outlier columns: the thread's 8 index elements are one
  // aligned 16-byte piece of the row
  u32x4 iw8 = {0, 0, 0, 0};
  const bool vec_idx = full && T == 16 && !d.inv_perm && S == 0 && (G & 7) == 0 &&
                       ((((uintptr_t)d.indices) & 15) == 0) && ((d.row_words & 3) == 0);
  if (vec_idx) {
    const int cb = j0 / G, g = j0 - cb * G;
    iw8 = *(const u32x4*)((const uint32_t*)d.indices + ((size_t)cb * d.num_indices + n) * d.row_words + (g >> 1));
  }
  // any other width, same conditions: two windows of 4 elements (nwin[k][0..3] = element 4k at bit 0)
  uint32_t nwin[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
  bool win_idx = full && T != 16 && !d.inv_perm && S == 0 && (G & 7) == 0;
  if (win_idx) {
    const int cb = j0 / G, g = j0 - cb * G;
    const uint32_t* row = (const uint32_t*)d.indices + ((size_t)cb * d.num_indices + n) * d.row_words;
    // a window that would run past the row end (the last chunk of a row): element by element instead
    const uint32_t bit1 = (uint32_t)(g + 4) * (uint32_t)T;
    win_idx = (int)(bit1 >> 5) + 5 <= d.row_words;
    if (win_idx) {
      const int g32 = (4 * T) & -(4 * T) & 31 ? ((4 * T) & -(4 * T)) : 32;
      const bool need5 = 4 * T > 96 + g32;   // some lane's window reaches a fifth word: a property of T
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const uint32_t bit = (uint32_t)(g + 4 * k) * (uint32_t)T;
        const uint32_t w0 = bit >> 5, off = bit & 31u;
        const dq_u32x4_a4 a = *(const dq_u32x4_a4*)(row + w0);
        const uint32_t w4 = need5 ? *(const dq_u32_a4*)(row + w0 + 4) : 0u;
        nwin[k][0] = __builtin_amdgcn_alignbit(a[1], a[0], off);
        nwin[k][1] = __builtin_amdgcn_alignbit(a[2], a[1], off);
        nwin[k][2] = __builtin_amdgcn_alignbit(a[3], a[2], off);
        nwin[k][3] = __builtin_amdgcn_alignbit(w4, a[3], off);
      }
    }
  }
  auto window_elem = [&](const uint32_t (&nw)[4], int p) -> uint32_t {   // p = e T: wave-uniform
    const int wi = p >> 5, sh = p & 31;
    const uint32_t lo = wi == 0 ? nw[0] : wi == 1 ? nw[1] : wi == 2 ? nw[2] : nw[3];
    const uint32_t hi = wi == 0 ? nw[1] : wi == 1 ? nw[2] : wi == 2 ? nw[3] : 0u;
    const uint32_t mask = T >= 32 ? 0xffffffffu : ((1u << T) - 1u);
    return __builtin_amdgcn_alignbit(hi, lo, (uint32_t)sh) & mask;
  };
  uint32_t w2[8][VP];  // [column][row pair]
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int j = j0 + q < I ? j0 + q : I - 1;  // ragged last chunk: recompute the last column
    const int c = d.inv_perm ? (int)d.inv_perm[j] : j;
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
        w2[q][p] = pr;
      }
    } else {
      const int cc = c - S;
      const int cb = cc / G, g = cc - cb * G;
      const uint32_t* row =
          (const uint32_t*)d.indices + ((size_t)cb * d.num_indices + n) * d.row_words;
      const uint32_t e = vec_idx ? ((iw8[q >> 1] >> (16 * (q & 1))) & 0xffffu)
                         : win_idx ? window_elem(nwin[q >> 2], (q & 3) * T) : unpack_elem(row, g, T);
      const uint32_t idx = e & ((1u << d.index_bits) - 1u);
      if constexpr (LDSTAB) {
        const uint32_t* cp = dq_tab + (size_t)idx * VP;
#pragma unroll
        for (int p = 0; p < VP; ++p) w2[q][p] = cp[p];
        if (d.res_bits) {
          const uint32_t ridx = (e >> d.index_bits) & ((1u << d.res_bits) - 1u);
          const uint32_t* rp = dq_tab + (size_t)(d.num_centroids + ridx) * VP;
#pragma unroll
          for (int p = 0; p < VP; ++p) w2[q][p] = DT::add2(w2[q][p], rp[p]);
        }
      } else {
        const uint32_t* cp = (const uint32_t*)d.centroids + ((size_t)cb * d.num_centroids + idx) * VP;
#pragma unroll
        for (int p = 0; p < VP; ++p) w2[q][p] = cp[p];
        if (d.res_bits) {
          const uint32_t ridx = (e >> d.index_bits) & ((1u << d.res_bits) - 1u);
          if constexpr (RESLDS) {
            const uint32_t* rp = dq_tab + (size_t)ridx * VP;
#pragma unroll
            for (int p = 0; p < VP; ++p) w2[q][p] = DT::add2(w2[q][p], rp[p]);
          } else {
            const uint32_t* rp =
                (const uint32_t*)d.res_centroids + ((size_t)cb * d.num_res_centroids + ridx) * VP;
#pragma unroll
            for (int p = 0; p < VP; ++p) w2[q][p] = DT::add2(w2[q][p], rp[p]);
          }
        }
      }
    }
    if (d.weight_scale) {
      const uint32_t s2 = vec_norm ? splat16((uint16_t)(sv8[q >> 1] >> (16 * (q & 1))))
                                   : splat16(((const uint16_t*)d.weight_scale)[j]);
      const uint32_t b2 = vec_norm ? splat16((uint16_t)(bv8[q >> 1] >> (16 * (q & 1))))
                                   : splat16(((const uint16_t*)d.weight_bias)[j]);
#pragma unroll
      for (int p = 0; p < VP; ++p) w2[q][p] = DT::add2(DT::mul2(w2[q][p], s2), b2);
    }
  }
#pragma unroll
  for (int p = 0; p < VP; ++p) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int o = n * V + 2 * p + h;
      if (o >= O) continue;
      // row o of the 8 columns: the low (h = 0) or high halves of w2[0..7][p]
      u32x4 r;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        r[k] = __builtin_amdgcn_perm(w2[2 * k + 1][p], w2[2 * k][p], h ? 0x07060302u : 0x05040100u);
      uint16_t* const dst = W + (size_t)o * I + j0;
      if (full && (((uintptr_t)W) & 15) == 0) {   // (rows are 2 I bytes, I % 8 == 0: W's alignment is every row's)
        *(u32x4*)dst = r;
      } else {
#pragma unroll
        for (int q = 0; q < 8; ++q)
          if (j0 + q < I) dst[q] = (uint16_t)(r[q >> 1] >> (16 * (q & 1)));
      }
    }
  }
}

template <typename DT, int V>
static hipError_t launch_v(const VptqLayerDesc& d, void* W, hipStream_t st) {
  const int col_blocks = (d.in_features + 2047) / 2048;
  const long long blocks = (long long)col_blocks * d.num_indices;
  if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
  const int tab_bytes = (d.num_centroids + d.num_res_centroids) * V * 2;
  const int res_bytes = d.num_res_centroids * V * 2;
  if (d.num_codebooks == 1 && tab_bytes <= kDqLdsMax)
    hipLaunchKernelGGL((dequant_kernel<DT, V, 1>), dim3((unsigned)blocks), dim3(256), tab_bytes, st, d,
                       (uint16_t*)W, col_blocks);
  else if (d.num_codebooks == 1 && res_bytes > 0 && res_bytes <= kDqLdsMax)
    hipLaunchKernelGGL((dequant_kernel<DT, V, 2>), dim3((unsigned)blocks), dim3(256), res_bytes, st, d,
                       (uint16_t*)W, col_blocks);
  else
    hipLaunchKernelGGL((dequant_kernel<DT, V, 0>), dim3((unsigned)blocks), dim3(256), 0, st, d, (uint16_t*)W,
                       col_blocks);
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
