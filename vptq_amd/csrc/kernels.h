// Internal launcher prototypes (one per .hip translation unit).
#pragma once
#include <atomic>
#include <hip/hip_runtime.h>

#include "../../include/vptq_hip.h"
#include "tune_env.h"

namespace vptq {

// gemv_generic.hip — every configuration
hipError_t launch_gemv_generic(const VptqLayerDesc& d, const void* x, void* y, int tokens,
                               bool out_f32, hipStream_t st);

// gemv_k256.hip — v=8, k=256 (+ kr=256), C=1, no outliers: LDS-resident,
// bank-conflict-free replicated codebooks.
bool gemv_k256_eligible(const VptqLayerDesc& d, int tokens);
const char* gemv_k256_name(const VptqLayerDesc& d, int tokens, int flags);
const char* gemv_k256_group_name(const VptqLayerDesc* descs, int n, int tokens, int flags);
hipError_t launch_gemv_k256(const VptqLayerDesc* descs, int n, const void* const* x,
                            void* const* y, int tokens, int flags, hipStream_t st);

// gemv_k256c.hip - the same format, one token, no permutation: ONE persistent launch that walks a
// chain of layers (next layer's codebook image, activations and index words requested while the current
// one streams).  n <= 32 layers of one dtype per launch.
bool gemv_k256c_eligible(const VptqLayerDesc& d, int tokens);
bool gemv_k256c_fills_device(const VptqLayerDesc* descs, int n, bool dependent);
bool gemv_k256c_exact_ok(const VptqLayerDesc& d, bool dependent);
bool gemv_k256c_selective_ok(const VptqLayerDesc& d, bool dependent);
size_t gemv_k256c_selective_bytes(const VptqLayerDesc* descs, int n);
// layers with an input permutation in an independent chain: x[perm] gathered into a workspace in front of the launch
size_t gemv_k256c_perm_bytes(const VptqLayerDesc& d);
hipError_t launch_permute_x(const VptqLayerDesc* descs, int n, const void* const* x, void* const* out, hipStream_t st);   // VPTQ_GEMV_EXACT inside the chain launch
hipError_t launch_gemv_k256c(const VptqLayerDesc* descs, int n, const void* const* x, void* const* y,
                             int flags, bool dependent, uint32_t* sync, hipStream_t st);

// gemv_gather.hip — v=8, k=65536 (+ residual 0 / 256 / 65536), C=1, no outliers:
// centroid rows gathered from L2.
bool gemv_gather_eligible(const VptqLayerDesc& d, int tokens);
hipError_t launch_gemv_gather(const VptqLayerDesc& d, const void* x, void* y, int tokens,
                              bool out_f32, hipStream_t st);

// gemv_gatherx.hip - every vector length, any codebook sizes (any total index width), several codebook
// groups, outlier columns of the same vector length: codebook rows gathered from L2 (what gemv_gather / gemv_lds do not take)
bool gemv_gatherx_eligible(const VptqLayerDesc& d, int tokens);
int gemv_gatherx_max_chunk(const VptqLayerDesc& d);   // token slots of one launch: 8 (v <= 8) or 4
hipError_t launch_gemv_gatherx(const VptqLayerDesc& d, const void* x, void* y, int tokens,
                               bool out_f32, hipStream_t st);

// gemv_lds.hip - v=8, one codebook, 256 < k <= 8192, kr <= 512: both codebooks LDS-resident,
// packed bit stream (T in {12, 13, 20, 21, 22}) or the v2 wire format
bool gemv_lds_eligible(const VptqLayerDesc& d, int tokens, int flags);
int gemv_lds_max_chunk(int dtype);
const char* gemv_lds_name(const VptqLayerDesc& d, int tokens, int flags);
// flags: VPTQ_GEMV_EXACT keeps one-token launches on the kernel with the reference's roundings
hipError_t launch_gemv_lds(const VptqLayerDesc& d, const void* x, void* y, int tokens, bool out_f32,
                           int flags, hipStream_t st);
bool gemv_lds_v2_eligible(const VptqV2Desc& d, int tokens);
hipError_t launch_gemv_lds_v2(const VptqV2Desc& d, const void* x, void* y, int tokens, bool out_f32,
                              int flags, hipStream_t st);

// gemv_sliced.hip - v8-k65536-0, one token, over the load-time derived sliced layout (LDS-local gathers)
// exact: the reference's roundings per weight (VPTQ_GEMV_EXACT) - scale and bias staged per column beside the activations
// (6 instead of 2 bytes of LDS per column: more slices for wide layers), one table only
bool gemv_sliced_eligible(const VptqLayerDesc& d, bool exact = false);
int gemv_sliced_slices(const VptqLayerDesc& d, bool exact = false);
int gemv_sliced_tables(const VptqLayerDesc& d);
int gemv_sliced_whole_table(const VptqLayerDesc& d, int table);   // VptqSlicedLayout::whole_table the layout of `table` must have   // layouts the layer needs: 1, or 2 (a residual codebook served as a second table)
size_t gemv_sliced_workspace_bytes(const VptqLayerDesc& d);
// gemv_hot.hip - VPTQ_GEMV_SELECTIVE over the sliced layouts: thresholds, x with the hot blocks zeroed, the hot blocks' exact products
bool gemv_hot_eligible(const VptqLayerDesc& d);
size_t gemv_hot_bytes(const VptqLayerDesc& d);
hipError_t launch_gemv_hot(const VptqLayerDesc& d, const void* x, void* extra, const void** x_masked, const float** corr, hipStream_t st);
hipError_t launch_gemv_sliced(const VptqLayerDesc& d, const VptqSlicedLayout* L, const void* x, void* y, int flags,
                              void* ws, hipStream_t st, const float* corr = nullptr);
// up to 3 layers of one format reading the same x (q / k / v, gate / up) in one launch
// gemv_sliced_tok.hip - 2 - 4 tokens over the same layouts (column windows of every list, phase by phase)
bool gemv_sliced_tok_eligible(const VptqLayerDesc& d, const VptqSlicedLayout* L, int tokens, bool exact = false);
int gemv_sliced_tok_one_pass_parts(const VptqLayerDesc& d, int tokens, bool exact);   // 0: column phases / not served; 1, 2, 4: one pass, that many window parts
size_t gemv_sliced_tok_workspace_bytes(const VptqLayerDesc& d, int tokens);
hipError_t launch_gemv_sliced_tok(const VptqLayerDesc& d, const VptqSlicedLayout* L, const void* x, void* y, int tokens, int flags,
                                  void* ws, hipStream_t st);
bool gemv_sliced_tok_groupable(const VptqLayerDesc* d, const VptqSlicedLayout* L, int n, int tokens, bool exact = false);
hipError_t launch_gemv_sliced_tok_group(const VptqLayerDesc* d, const VptqSlicedLayout* L, int n, const void* x, void* const* y,
                                        int tokens, int flags, void* const* ws, hipStream_t st);
bool gemv_sliced_groupable(const VptqLayerDesc* d, int n, bool exact = false);
hipError_t launch_gemv_sliced_group(const VptqLayerDesc* d, const VptqSlicedLayout* L, int n, const void* x, void* const* y,
                                    int flags, void* const* ws, hipStream_t st, int tokens = 1, const float* corr = nullptr);
// (VPTQ_GEMV_EXACT) 2 / 3 tokens in ONE pass of the one-token kernel: x [tokens][in], y[i] [tokens][out], ws[i]: accumulator words
bool gemv_sliced_exact_tokens_ok(const VptqLayerDesc& d, int tokens);
int gemv_sliced_exact_tokens_parts(const VptqLayerDesc& d, int tokens);   // 0: not served; 1: all columns staged; 2 / 4: window parts (needs wstart)
size_t gemv_sliced_exact_tokens_workspace_bytes(const VptqLayerDesc& d, int tokens);
// gemm_k256t.hip - canonical format, fp16 / bf16, up to 16 tokens in one pass over the indices (transposing
// gather -> 16x16x32 MFMA with tokens as M; folded arithmetic; needs a workspace for the operand-ordered activations)
bool gemm_k256t_eligible(const VptqLayerDesc& d, int tokens, int flags);
size_t gemm_k256t_workspace_bytes(const VptqLayerDesc& d);
hipError_t launch_gemm_k256t(const VptqLayerDesc& d, const void* x, void* y, int tokens, bool out_f32, void* ws,
                             hipStream_t st);
// gemm_k256.hip - canonical format, fp16, up to 16 tokens in one launch (tokens = MFMA M)
bool gemm_k256_eligible(const VptqLayerDesc& d, int tokens, int flags);
hipError_t launch_gemm_k256(const VptqLayerDesc& d, const void* x, void* y, int tokens, bool out_f32,
                            hipStream_t st);

// gemm_fused.hip - canonical format, many tokens: dequantised tile -> LDS -> 32x32x16 MFMA
bool gemm_fused_eligible(const VptqLayerDesc& d);
size_t gemm_fused_workspace_bytes(const VptqLayerDesc& d, int tokens);
hipError_t launch_gemm_fused(const VptqLayerDesc& d, const void* x, void* y, int tokens, void* workspace,
                             size_t workspace_bytes, hipStream_t st);

// dequant.hip
hipError_t launch_dequant(const VptqLayerDesc& d, void* W, hipStream_t st);

// gemv_v2.hip
hipError_t launch_gemv_v2(const VptqV2Desc& d, const void* x, void* y, int tokens,
                          bool out_f32, hipStream_t st);

}  // namespace vptq
