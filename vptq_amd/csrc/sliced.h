// Shared by gemv_sliced.hip (one token) and gemv_sliced_tok.hip (2 - 4 tokens): launch constants, the parameter block of
// one layer and the host-side helpers that say what a layer's layout looks like.
#pragma once
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace vptq {

constexpr int kSLThreads = 1024;
constexpr int kSLWaves = kSLThreads / 64;
// 8 slices of 8192 entries (128 KiB of LDS) while the staged activations fit beside them, else 16 slices of 4096
// (64 KiB): 8 slices hold f16(s x) of 14336 columns (14080 with the 4 KiB residual codebook), 16 slices of 32768
constexpr int kSLMaxG8 = 14336, kSLMaxG8Res = 14080, kSLMaxG16 = 32768;
constexpr uint32_t kSLLdsLimit = 163840;
constexpr int kSLMaxRowsPerWave = 64;                    // (their block counts sit in the lanes of one register)

struct SlicedParams {
  const uint32_t* elems;
  const uint8_t* res;       // residual index per element (same order and padding), or null
  const uint32_t* rcent;    // [256][8] halves, or null
  const int32_t* blocks;    // [8][N]
  const int32_t* first;     // [8][N]
  const uint32_t* cent;     // [65536][8] halves
  const uint16_t* x;
  const uint16_t* scale;
  const uint16_t* wbias;    // input-feature order (folded arithmetic: sum b x)
  const uint16_t* cbias;    // COLUMN order (reference roundings: w = f16(f16(u * s) + b) per weight): bias_permuted or weight_bias
  const uint16_t* perm;     // column c of the quantised matrix multiplies input feature perm[c]; `scale` is then in column order
  const uint16_t* bias;
  const float* corr;        // [N x v] float32 added to the sums before the one rounding, or null: the hot blocks' exact products of
                            // VPTQ_GEMV_SELECTIVE (gemv_hot.hip), whose activations this launch reads as zeros
  // TWO tables in one launch (65536 residual centroids): the residual table's layout and codebook; its workgroups are
  // "slices" NSL .. 2 NSL - 1 of the same row blocks
  const uint32_t* elems2;
  const int32_t* blocks2;
  const int32_t* first2;
  const uint32_t* cent2;
  // what a workgroup of table t copies into LDS: tab_t bytes from cent + slice x stride_t (stride = tab: its slice of
  // the table, element words carry the index inside the slice; stride = 0: the WHOLE table - small residual tables -,
  // element words carry the full index); the staged activations start at x_off >= max(tab)
  uint32_t tab0, tab1, stride0, stride1, x_off;
  float* partial;           // [slices][N * 8]
  uint32_t* arrived;        // [row blocks] workgroups of the row block that have stored their partial sums (0 between launches)
  void* y;
  int N, G, O, rows_per_wave, n_rowblocks, out_f32;
  // several tokens in one pass (gemv_sliced.hip, TOK > 1): elements between two tokens of x / y, words between their accumulators
  int x_stride, y_stride, acc_stride;
  // ... in WINDOW PARTS (WPT): the layout's window table [slices][N][windows + 1], workgroups per (slice, row block) (2 or 4), columns
  // per layout window, columns the LDS map is laid out for (the widest part)
  const int32_t* wstart;
  int wparts, wcols, wstage;
};
constexpr int kSLWindows = VPTQ_SLICED_WINDOWS;

// f(slot 0), ... f(slot kSLQueue - 1) with the slot as a compile-time constant
template <int I0, int I1, typename F>
static __device__ __forceinline__ void sl_for_range(F&& f) {
  if constexpr (I1 - I0 == 1) f(std::integral_constant<int, I0>{});
  else {
    sl_for_range<I0, (I0 + I1) / 2>(f);
    sl_for_range<(I0 + I1) / 2, I1>(f);
  }
}
template <int Q, typename F>
static __device__ __forceinline__ void sl_for_slots(F&& f) { sl_for_range<0, Q>(f); }

// ---- host side (gemv_sliced.hip)
bool sl_res256(const VptqLayerDesc& d);   // v = 8 with the 256-entry residual table: a byte per element beside the main layout
bool sl_two(const VptqLayerDesc& d);      // any other residual table: a second table with a layout of its own
uint32_t sl_tab_bytes(const VptqLayerDesc& d, int k, int whole, bool exact = false);   // bytes a workgroup of a k-entry table holds
bool sl_layout_ok(const VptqLayerDesc& d, const VptqSlicedLayout& L, int nsl, bool res, int k);

}  // namespace vptq
