/*
 * vptq_hip.h — C ABI of libvptq_hip.so: the MI355X (gfx950) replacement for the
 * native extension behind microsoft/VPTQ's `vptq.ops` hot path.
 *
 * Drop-in boundary.  The reference binds three C++/CUDA entry points through
 * pybind11 (csrc/ops.cc:44-55); each function below replaces one of them:
 *
 *   vptq_quant_gemv     <- `quant_gemv`    = wquant_act16_gemv     csrc/ops.cc:20-30,
 *                                            csrc/quant_gemv.cu:241-294
 *   vptq_dequant        <- `dequant`                                csrc/ops.cc:9-18,
 *                                            csrc/dequant.cu:227-287
 *   vptq_quant_gemv_v2  <- `quant_gemv_v2`                          csrc/ops.cc:32-38,
 *                                            csrc/quant_gemv_v2.cu:25-180
 *
 * Plain C: raw device pointers + sizes, no torch / pybind / C++ types.
 *
 * Ownership ... the caller owns every buffer (inputs, outputs, workspace); the
 *               library allocates nothing and keeps no per-call state.  The
 *               reference ops allocate their own output (quant_gemv.cu:206).
 * Streams ..... every launch goes to the hipStream_t passed as `stream`
 *               (NULL = the default stream).  Asynchronous, no host sync, no
 *               host read of device data => hipGraph-capturable.  The reference
 *               uses at::cuda::getCurrentCUDAStream() (quant_gemv.cu:180).
 * Errors ...... return 0 on success; < 0 = VPTQ_E_* validation error (nothing
 *               launched); > 0 = hipError_t from the launch.  A message is kept
 *               per thread in vptq_last_error().  The reference throws
 *               c10::Error via TORCH_CHECK (csrc/util/common.h:11-19).
 * Threading ... re-entrant; no globals besides one-time kernel attribute setup.
 */
#ifndef VPTQ_HIP_H
#define VPTQ_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VPTQ_ABI_VERSION 10

#if defined(__GNUC__)
#define VPTQ_API __attribute__((visibility("default")))
#else
#define VPTQ_API
#endif

/* element type of activations / codebooks / scales / outputs */
enum { VPTQ_DTYPE_F16 = 0, VPTQ_DTYPE_BF16 = 1 };

/* error codes (negative) */
enum {
  VPTQ_OK = 0,
  VPTQ_E_NULL = -1,        /* required pointer is NULL            */
  VPTQ_E_SHAPE = -2,       /* inconsistent shapes / sizes         */
  VPTQ_E_UNSUPPORTED = -3, /* valid but no kernel for this combo  */
  VPTQ_E_ALIGN = -4,       /* pointer not aligned as required     */
  VPTQ_E_TOKENS = -5,      /* tokens outside [1, VPTQ_GEMV_MAX_TOKENS] */
  VPTQ_E_WORKSPACE = -6    /* workspace too small                 */
};

/* flags for vptq_quant_gemv / vptq_quant_gemv_v2 */
enum {
  /* Default arithmetic of the fused GEMV (ABI >= 3), where a kernel implements it (the
   * canonical v=8 / 256+256 format: fp16 with 1-2 tokens on any launch and with 3-4 tokens on
   * launches of >= 576 vector-rows, bf16 with 1-4 tokens on launches of >= 128 vector-rows;
   * the LDS-resident formats v=8, 256 < k <= 8192, kr <= 512: one-token calls on layers of
   * >= 1024 vector-rows, bf16 always):
   * the folded form
   *   y = sum_g (c + r) * (scale_g * x_g) + sum_g bias_g * x_g + bias,   fp32 accumulate,
   * inside the parity bar of <= 1e-3 (max-normalised) against the reference CPU path, but
   * its weights are not rounded to 16 bits step by step.  FAST_MATH asks for it explicitly
   * (what ABI 2 required); it is accepted and changes nothing. */
  VPTQ_GEMV_FAST_MATH = 1 << 0,
  /* force the generic kernel (testing / A-B) */
  VPTQ_GEMV_FORCE_GENERIC = 1 << 1,
  /* Reproduce the reference CPU path's roundings: every weight is rebuilt as
   * r16(r16(r16(c+r)*scale)+bias) in the 16-bit type (bit-identical to vptq_dequant and to
   * the reference's torch path), then x*w is accumulated in fp32 and rounded once.  All
   * other kernels (other formats, small layers with 3+ tokens or bf16) always work this way;
   * bf16 layers of the LDS-resident formats are routed to the L2-gather kernel for it. */
  VPTQ_GEMV_EXACT = 1 << 2,
  /* canonical format only: pick the persistent MFMA kernel wherever it is instantiated /
   * never pick it (testing / A-B; the default chooses by launch size) */
  VPTQ_GEMV_FORCE_MFMA = 1 << 3,
  VPTQ_GEMV_FORCE_VALU = 1 << 4,
  /* y is float32 [tokens, O]: the fp32 sums (output bias included when desc->bias is set) are
   * stored un-rounded.  For callers that combine several launches before the one rounding of
   * the reference's F.linear - the partial outputs of a row-parallel (input-column) shard,
   * summed by an all-reduce.  Every kernel honours it (ABI >= 4). */
  VPTQ_GEMV_OUT_F32 = 1 << 5,
  /* vptq_quant_gemv_chain only: layer i + 1 reads what layer i wrote (x[i + 1] aliases y[i]);
   * without it the layers of a chain must be independent of each other (ABI >= 6) */
  VPTQ_GEMV_CHAIN_DEPENDENT = 1 << 6,
  /* take the one-pass batched-decode kernel (gemm_k256t: canonical format, 1-16 tokens per launch,
   * needs the workspace) wherever it is eligible, not only where it is the fastest (bf16, 5+
   * tokens): testing / A-B */
  VPTQ_GEMV_FORCE_BATCHED = 1 << 7,
  /* vptq_quant_gemv_sliced_grouped (ABI >= 9): the n descriptors are COLUMN PARTS of one layer - see there */
  VPTQ_GEMV_COLUMN_PARTS = 1 << 8,
  /* SELECTIVE roundings (ABI >= 10): the folded form, with the reference's three roundings per weight (VPTQ_GEMV_EXACT's
   * arithmetic) on the activation columns that dominate the token - |f16(scale_g x_g)| >= 6 x the rms of f16(scale x) over the
   * layer's columns.  The folded form's distance to the reference is a sum of per-column rounding errors that average out over
   * thousands of columns when the activations are dense and do not when a handful of channels carry the token (massive
   * activations): the blocks of 128 columns that hold such a column are rebuilt bit-exactly, the rest stays folded.  Kernels
   * that implement it (fp16 and bf16): the persistent chain launch (independent layers; needs the workspace
   * vptq_quant_gemv_chain_workspace_bytes_for(descs, n, flags) asks for - 16 bytes per layer + 4 per output: thresholds and
   * the hot blocks' exact products) and the persistent MFMA kernel of one-layer / grouped launches (1 token, up to
   * 14336 columns, at most 16 row groups per workgroup; thresholds over the 512 columns a wave stages, then the 16 most dominant blocks);
   * every other kernel / layer / token count takes VPTQ_GEMV_EXACT instead (always at least as close to the reference).
   * With VPTQ_GEMV_EXACT set as well, EXACT wins.  Not bit-equivalent (55 - 65 % of the outputs bit-identical); counted on
   * checkpoint-like layers: 2 of 12 300 above the 1e-3 bar at 1.00e-3 / 1.09e-3 through the chain launch (folded form: 30
   * of 4100; the reference's roundings: 0 by construction) - an opt-in, not the default. */
  VPTQ_GEMV_SELECTIVE = 1 << 9
};

/* most tokens vptq_quant_gemv accepts (fp16 layers of the canonical format; every other layer:
 * VPTQ_GEMV_MAX_TOKENS_ANY); whether the fused path is also the FASTER one for a layer at a
 * token count is what vptq_quant_gemv_max_tokens() answers */
#define VPTQ_GEMV_MAX_TOKENS 64
#define VPTQ_GEMV_MAX_TOKENS_ANY 16

/*
 * One VQuantLinear layer in the reference's on-disk tensor formats
 * (vptq/layers/vqlinear.py:97-240).  All pointers are DEVICE pointers; nullable
 * ones are marked.  I = outlier_size + num_codebooks * group_size.
 *
 *   Wq[n*v+t, S + cb*G + g] = centroids[cb, idx, t] + res_centroids[cb, ridx, t]
 *   Wq[m*ov+t, g]           = outlier_centroids[0, outlier_indices[0,m,g], t]   (g < S)
 *   W[o, j] = Wq[o, inv_perm[j]] * weight_scale[j] + weight_bias[j]
 *   y[., o] = sum_j x[., j] * W[o, j] + bias[o]
 *
 * indices: int32 [C, N, row_words]; each (cb, n) row is a little-endian bit
 * stream, element g = bits [g*T, (g+1)*T), T = index_bits + res_bits,
 * value = (ridx << index_bits) | idx   (vptq/utils/pack.py:26-67).
 */
typedef struct VptqLayerDesc {
  int32_t in_features;           /* I */
  int32_t out_features;          /* O */
  int32_t vector_len;            /* v: even, 2..16 */
  int32_t num_codebooks;         /* C = group_num */
  int32_t group_size;            /* G */
  int32_t num_centroids;         /* k (power of two, <= 65536) */
  int32_t num_res_centroids;     /* kr; 0 = no residual codebook */
  int32_t index_bits;            /* log2 k  */
  int32_t res_bits;              /* log2 kr */
  int32_t row_words;             /* ceil(G*T/32) */
  int32_t num_indices;           /* N = ceil(O / v) */
  int32_t outlier_size;          /* S; 0 = no outliers */
  int32_t outlier_vector_len;    /* ov */
  int32_t num_outlier_centroids; /* ko */
  int32_t num_outlier_indices;   /* M = ceil(O / ov) */
  int32_t dtype;                 /* VPTQ_DTYPE_* */

  const int32_t* indices;          /* [C, N, row_words]                        */
  const void* centroids;           /* [C, k, v]                                 */
  const void* res_centroids;       /* [C, kr, v]            NULL iff kr == 0    */
  const uint16_t* outlier_indices; /* [1, M, S]             NULL iff S == 0     */
  const void* outlier_centroids;   /* [1, ko, ov]           NULL iff S == 0     */
  const uint16_t* perm;            /* [I] uint16            NULL = identity     */
  const uint16_t* inv_perm;        /* [I] argsort(perm)     needed by dequant when perm != NULL */
  const void* weight_scale;        /* [I]   NULL (with weight_bias) = no norm   */
  const void* weight_bias;         /* [I]                                       */
  const void* bias;                /* [O]                   NULL = none         */
  /* Optional derived state (all nullable; results never depend on them): */
  const void* scale_permuted;      /* [I] weight_scale[perm[c]]: lets the GEMV read scale in  */
  const void* bias_permuted;       /* [I] weight_bias[perm[c]]   column order (perm != NULL)  */
  const void* prefetch;            /* read-only range the GEMV may touch to warm L2 / Infinity  */
  int64_t prefetch_bytes;          /* Cache for the NEXT launch (e.g. the next layer's indices) */
} VptqLayerDesc;

/*
 * v2 wire format (tests/test_quant_gemv.py:49-109, csrc/quant_gemv_v2.cu:25-180):
 * UNPACKED indices laid out [N][I]; one codebook; residual ids uint8 or uint16.
 *   W^T[i, n*v+t] = centroids[ids[n*I+i], t] + res_centroids[rids[n*I+i], t]
 *   W^T = scale[i] * W^T + sbias[i];  y = x @ W^T + bias
 */
typedef struct VptqV2Desc {
  int32_t in_features;
  int32_t out_features;
  int32_t vector_len;
  int32_t num_centroids;
  int32_t num_res_centroids; /* 0 = none */
  int32_t res_index_bytes;   /* 1 (uint8) or 2 (uint16) */
  int32_t dtype;
  int32_t reserved;
  const uint16_t* indices;     /* [N * I] */
  const void* centroids;       /* [1, k, v] */
  const void* res_indices;     /* [N * I] uint8/uint16, NULL iff kr == 0 */
  const void* res_centroids;   /* [1, kr, v] */
  const void* scale_weights;   /* [I] nullable */
  const void* scale_bias;      /* [I] nullable */
  const void* bias;            /* [O] nullable */
} VptqV2Desc;

VPTQ_API int vptq_abi_version(void);

/* thread-local message for the last non-zero return on this thread */
VPTQ_API const char* vptq_last_error(void);

/*
 * y[tokens, O] = x[tokens, I] @ W^T + bias, fused dequant, tokens in
 * [1, VPTQ_GEMV_MAX_TOKENS].  x, y dense row-major in desc->dtype.
 * workspace: optional, never required.  vptq_quant_gemv_workspace_bytes() tells how many bytes
 * (16-byte aligned, device memory, contents don't matter) let the call take its fastest kernel: for
 * 2+ tokens of the canonical 256 + 256 format that is gemm_k256t_kernel (up to 16 tokens in ONE pass
 * over the indices, fp16 and bf16; its pre-pass writes the operand-ordered activations there).
 * With NULL / fewer bytes the call uses the kernels that need none (same results within the
 * parity bar).  Launches of one call and calls on one stream may share a workspace.
 */
VPTQ_API int vptq_quant_gemv(const VptqLayerDesc* desc, const void* x, void* y, int tokens,
                    int flags, void* workspace, size_t workspace_bytes, void* stream);

VPTQ_API size_t vptq_quant_gemv_workspace_bytes(const VptqLayerDesc* desc, int tokens, int flags);

/*
 * Largest token count for which vptq_quant_gemv is the path to take for this layer; above it
 * vptq_dequant + a dense GEMM is faster (the reference switches at 3 for every format,
 * vptq/ops/quant_gemm.py:213).  48 for fp16 layers of the canonical v=8 / 256+256 format (5+
 * tokens = launches of the batched-decode kernel, 16 tokens each: 18 us per 8192^2 layer and
 * launch against 76-80 us for dequant + GEMM; measured crossover 48-64 tokens,
 * tools/tokens_crossover.py), 16 for its bf16 layers (launches of <= 4 tokens), 8 for every
 * other format.  0 if desc is invalid.
 */
VPTQ_API int vptq_quant_gemv_max_tokens(const VptqLayerDesc* desc);

/*
 * n independent layers in ONE launch (q/k/v or gate/up projections of a decoder
 * layer, or any batch of VQuantLinear GEMVs).  descs is a HOST array; x[i], y[i]
 * are device pointers for layer i.  Every layer must be eligible for the same
 * kernel family (checked); n <= VPTQ_GROUP_MAX.
 */
#define VPTQ_GROUP_MAX 64
VPTQ_API int vptq_quant_gemv_grouped(const VptqLayerDesc* descs, int n, const void* const* x,
                            void* const* y, int tokens, int flags, void* stream);

/*
 * n layers walked ONE AFTER THE OTHER by a single persistent launch (ABI >= 6): what the reference
 * does as n calls of `quant_gemv` (vptq/ops/quant_gemm.py:214-228, one per VQuantLinear of a decode
 * step), each of which pays a launch boundary, a prologue (codebooks into shared memory,
 * activations) and an epilogue.  Here every workgroup streams layer i's index words and, while it
 * does, requests layer i + 1's codebooks, activations and first index words, so HBM never idles
 * between layers.  Results are those of n vptq_quant_gemv calls in order (same arithmetic).
 *   - independent layers (default): no layer reads another layer's output (q / k / v, gate / up,
 *     a ring of benchmark layers): the layers overlap freely.
 *   - VPTQ_GEMV_CHAIN_DEPENDENT: x[i + 1] may be y[i]; a device-scope arrival counter per layer
 *     orders them.  Needs workspace of vptq_quant_gemv_chain_workspace_bytes(n, flags) bytes (the
 *     call clears it on the stream).
 * Served by one persistent launch per <= 32 layers when every layer is of the canonical v=8 / 256+256 format,
 * one dtype, without a permutation (absorb it first), tokens == 1 and the list has more than 8 layers
 * (vptq_quant_gemv_chain_kernel_name says "gemv_k256c_kernel"; with VPTQ_GEMV_EXACT: fp16, independent layers).
 * Shorter lists of independent layers - and lists the persistent kernel does not take - go out as ONE
 * vptq_quant_gemv_grouped launch ("grouped": faster than the persistent launch up to 8 layers; it serves members it
 * has no common kernel for one by one); a single layer, mixed dtypes and dependent lists the persistent kernel does
 * not take are executed as n vptq_quant_gemv calls ("per-layer").  descs is a HOST array; x[i], y[i] device pointers.
 */
#define VPTQ_CHAIN_MAX 1024
VPTQ_API int vptq_quant_gemv_chain(const VptqLayerDesc* descs, int n, const void* const* x,
                          void* const* y, int tokens, int flags, void* workspace,
                          size_t workspace_bytes, void* stream);
VPTQ_API size_t vptq_quant_gemv_chain_workspace_bytes(int n, int flags);
/* the same + (independent lists, ABI >= 7) room for x[perm] of every layer with an input permutation: with that much
 * 256-byte aligned workspace such layers stay in the persistent launch (x is gathered by one small launch in front of
 * it); with less they are served by grouped / single launches, which apply permutations themselves.
 * flags | VPTQ_GEMV_SELECTIVE (ABI >= 10): + per launch of <= 32 layers 16 bytes per layer and 4 per output, in front of the
 * rest; with less (or NULL) the call takes VPTQ_GEMV_EXACT for every layer */
VPTQ_API size_t vptq_quant_gemv_chain_workspace_bytes_for(const VptqLayerDesc* descs, int n, int flags);
VPTQ_API const char* vptq_quant_gemv_chain_kernel_name(const VptqLayerDesc* descs, int n, int tokens,
                                              int flags);

/*
 * Many tokens (prefill): y[tokens, O] = x[tokens, I] @ W^T + bias with the dequantisation FUSED
 * into the GEMM - replaces `dequant` + F.linear of the reference (vptq/ops/quant_gemm.py:231-274),
 * which materialises the dense W (2 O I bytes) on every call.  Canonical v=8 / 256+256 format
 * without a permutation (absorb it first); fp16: the tile holds the reference's bits
 * r16(r16(r16(c+r)*s)+b), fp32 accumulate; bf16: folded form.  bf16 needs a workspace of
 * vptq_quant_gemm_workspace_bytes() (tokens floats).  vptq_quant_gemm_supported() = 1 when a
 * fused kernel exists for the layer; otherwise use vptq_dequant + a dense GEMM.
 */
VPTQ_API int vptq_quant_gemm_supported(const VptqLayerDesc* desc);
VPTQ_API size_t vptq_quant_gemm_workspace_bytes(const VptqLayerDesc* desc, int tokens);
VPTQ_API int vptq_quant_gemm(const VptqLayerDesc* desc, const void* x, void* y, int tokens, int flags,
                    void* workspace, size_t workspace_bytes, void* stream);

/*
 * One token over a LOAD-TIME DERIVED LAYOUT of a large-codebook layer (k = 65536 main centroids; v = 8 with residual none,
 * 256 or 65536: "v8-k65536-0" / "-256" / "-65536"; v = 16 with residual none or 65536: "v16-k65536-0" / "-65536" - the
 * formats of most published checkpoints; ABI >= 6, the 65536-residual and v = 16 formats since round 4).  The reference
 * gathers centroid rows from a 1 - 2 MiB codebook through the caches (csrc/kernels/quant_gemv.cuh:11-186); here every row's
 * elements are bucketed ONCE per layer by the top bits of their index, so that a workgroup holds its slice of the codebook
 * in LDS:
 *   elems  : uint32, for slice s = 0..S-1 (E = 65536 / S entries per slice), for row n = 0..N-1
 *            (N = desc->num_indices): the elements of row n whose index / E == s, in any order, padded to a
 *            multiple of 64 with the word (column = group_size, local = 0);
 *            element word = column | (index mod E) << 16
 *   blocks : int32 [S][N], 64-element blocks of (s, n);  first : int32 [S][N], index of its first block
 *            (prefix sum of `blocks` in (s, n) order)
 *   rows_per_wave : 1 .. 64 consecutive rows per wave (16 waves per workgroup); the layout does not
 *            depend on it
 * (vptq_amd/utils/sliced.py builds it with torch.)  It costs 2x (residual 256: 1.7x) the packed indices in device memory on
 * top of them; the state-dict tensors are untouched.  A layer with 65536 RESIDUAL centroids is served table by table
 * - (c + r) s x = c s x + r s x: the residual table's (slice, row block) workgroups run beside the main table's in the same
 * launch - and `layout` then points to TWO consecutive structs with the same rows_per_wave: [0] built from the main
 * indices, [1] from the residual indices (`res` unused).  workspace: vptq_quant_gemv_sliced_workspace_bytes, 16-byte aligned,
 * ZERO-FILLED ONCE by the caller before its first use - one 64-bit accumulator word per output (fixed-point sum | arrivals)
 * that the slices' partial sums are added to with returning atomics; the last arriver of an output rounds, stores y and
 * puts the word back to zero: every call leaves the workspace zero.  One workspace per layer and STREAM (two calls in flight
 * on different streams must not share one).  Arithmetic: flags = 0 the folded form (parity bar for dense activations, not
 * bit-equivalent); VPTQ_GEMV_EXACT (ABI >= 8) the reference's roundings per weight over a layout with the slice count
 * vptq_sliced_layout_supported_for(desc, VPTQ_GEMV_EXACT) answers.
 * Layers this path takes: group_size <= 32768; a permutation is applied while the activations are staged.
 */
typedef struct VptqSlicedLayout {
  const void* elems;
  const void* blocks;
  const void* first;
  const void* res;          /* uint8 per element (same order, padding = 0): residual index of v = 8's 256-entry table; NULL without
                             * residual codebook.  VPTQ_GEMV_EXACT layouts (ABI >= 8) of layers with ANY OTHER residual codebook:
                             * uint16 per element - the residual entry is gathered from the codebook in device memory */
  int32_t rows_per_wave;
  int32_t elems_per_lane;   /* 1 (or 0): a block = 64 elements */
  int32_t n_slices;         /* 8 (or 0) / 16 / 32: what vptq_sliced_layout_supported() answers for the layer */
  int32_t whole_table;      /* 0: element words carry the index INSIDE the slice (the workgroup holds its slice of the table);
                             * 1: the full index (every workgroup of this table holds all of it: small residual tables) */
  const void* wstart;       /* int32 [S][N][VPTQ_SLICED_WINDOWS + 1], or NULL (one token only): every (s, n) list is ordered by
                             * COLUMN WINDOW - window w = columns [w C, (w + 1) C), C = group_size / 4 rounded up to a multiple
                             * of 8, the last window taking what is left - and wstart[s][n][w] is the position inside the list
                             * at which window w begins, [VPTQ_SLICED_WINDOWS] the list's length without padding.  What
                             * vptq_quant_gemv_sliced_tokens walks; one token ignores it */
} VptqSlicedLayout;
#define VPTQ_SLICED_WINDOWS 4
/* 0 = not a layer of this path; else the number of slices its layout(s) must have: v = 8: 8 slices of 8192 entries while
 * the activations fit in LDS beside them (group_size <= 14336, 14080 with the 256-entry residual codebook), else 16 of
 * 4096; v = 16 (32-byte entries): 16 slices of 4096 entries, beyond 14336 columns 32 of 2048 */
VPTQ_API int vptq_sliced_layout_supported(const VptqLayerDesc* desc);
/* ... and for the arithmetic a call will ask for (ABI >= 8): flags = 0 as above; VPTQ_GEMV_EXACT - the reference's roundings per
 * weight, w = f16(f16(f16(c + r) * s) + b) (vptq/ops/quant_gemm.py:121,155-156), what gemv_gather computes through the caches -
 * needs c and r in one lane, so it always takes ONE layout, bucketed by the main index (vptq_sliced_layout_tables() is the folded
 * form's answer): no residual codebook; v = 8 with 256 residual centroids (`res` bytes, table in LDS); any other residual
 * codebook: `res` = uint16 residual indices, the entry gathered from device memory (one of the gather kernel's two cache
 * gathers per element).  Scale, bias and x of every column sit beside the slice, 6 instead of 2 bytes of LDS per column:
 * v = 8: 8 slices up to 5376 columns (4704 with the 256-entry table), 16 up to 16288 (15616); v = 16: 16 / 32; wider layers: 0.
 * A layout built with THAT slice count serves vptq_quant_gemv_sliced(..., flags | VPTQ_GEMV_EXACT, ...) and - one-table formats,
 * ABI >= 9 - vptq_quant_gemv_sliced_tokens(..., flags | VPTQ_GEMV_EXACT, ...). */
VPTQ_API int vptq_sliced_layout_supported_for(const VptqLayerDesc* desc, int flags);
/* 0, or how many consecutive VptqSlicedLayout structs vptq_quant_gemv_sliced takes for this layer: 1 (no residual codebook;
 * v = 8 with 256 residual centroids: `res` bytes), 2 (any other residual codebook: a second table with a layout of its own) */
VPTQ_API int vptq_sliced_layout_tables(const VptqLayerDesc* desc);
/* the `whole_table` value the layout of table 0 / 1 must be built with (1: a small residual table every workgroup holds whole) */
VPTQ_API int vptq_sliced_layout_whole_table(const VptqLayerDesc* desc, int table);
VPTQ_API size_t vptq_quant_gemv_sliced_workspace_bytes(const VptqLayerDesc* desc);
VPTQ_API int vptq_quant_gemv_sliced(const VptqLayerDesc* desc, const VptqSlicedLayout* layout, const void* x,
                           void* y, int flags, void* workspace, size_t workspace_bytes, void* stream);
/* flags | VPTQ_GEMV_SELECTIVE (ABI >= 10; the FOLDED layouts, i.e. without VPTQ_GEMV_EXACT): the folded form over the layouts, with the
 * reference's roundings on the blocks of 128 columns an activation dominates.  A small launch in front (gemv_hot.hip) finds the
 * threshold, writes the activation with those blocks' input features zeroed - what the folded launch then reads - and the blocks'
 * exact products per output (entries gathered from the codebooks in device memory: few columns), which the folded launch adds before
 * its one rounding.  fp16 layers with scale and bias (vptq_quant_gemv_sliced_selective_supported); workspace:
 * vptq_quant_gemv_sliced_workspace_bytes_for(desc, flags) bytes, 256-byte aligned, zero-filled once (the accumulator words come first).
 * What it buys: the two-table formats (v8-k65536-65536, v16-k65536-65536) at the folded form's speed - 8192^2: 59 / 46 us in the
 * reference's roundings, ~21 / ~22 here - without the folded form's activation-dependent failures; opt-in, like every use of the flag. */
VPTQ_API int vptq_quant_gemv_sliced_selective_supported(const VptqLayerDesc* desc);
VPTQ_API size_t vptq_quant_gemv_sliced_workspace_bytes_for(const VptqLayerDesc* desc, int flags);

/* 2 - 4 tokens (5 - 8 where 16 bytes of activations per column still fit the LDS in 4 phases: layers of up to ~4600 columns)
 * over the same layouts (ABI >= 7; needs `wstart`): x [tokens][in_features], y [tokens][out_features]
 * (float32 with VPTQ_GEMV_OUT_F32), one launch.  The activations of T tokens do not fit beside a slice, so the kernel takes
 * the columns in 1, 2 or 4 phases (as few as the LDS allows: 4096-column layers need one for 2 - 3 tokens) and walks the
 * phase's column windows of every list.  workspace: vptq_quant_gemv_sliced_tokens_workspace_bytes(desc, tokens), zero-filled
 * once, one per layer and stream, not shared with the one-token call; a workspace for T tokens serves fewer as well (the
 * arrival counters sit in front).  VPTQ_E_UNSUPPORTED where the layer, the layout
 * (no wstart) or the token count is not served: take vptq_quant_gemv.  Replaces the same reference kernel for
 * 1 < tokens < 16 (vptq/ops/quant_gemm.py:213, csrc/kernels/quant_gemv.cuh:11-186).
 * flags | VPTQ_GEMV_EXACT (ABI >= 9): the reference's roundings per weight over an EXACT layout (vptq_sliced_layout_supported_for(desc,
 * VPTQ_GEMV_EXACT) slices) of a one-table format (no residual codebook; v = 8 with 256 residual centroids): scale and bias of a
 * column are staged beside its activations (8 more bytes of LDS per column and phase), every weight is rebuilt as
 * f16(f16(f16(c + r) s) + b) in the matrix pipe's operand layout; vptq_quant_gemv_sliced_tokens_supported_for says whether the
 * layer's columns fit (4 phases x (2 x token slots + 8) bytes beside the slice: 5 - 8 tokens up to ~16000 columns).  2 / 3 tokens (v = 16:
 * 2) of a layer whose slice leaves room for 2 x tokens + 4 bytes per column take ONE PASS of the one-token kernel instead (no `wstart`
 * needed; accumulator words behind the arrival counters of the same workspace). */
VPTQ_API int vptq_quant_gemv_sliced_tokens_supported(const VptqLayerDesc* desc, const VptqSlicedLayout* layout, int tokens);
VPTQ_API int vptq_quant_gemv_sliced_tokens_supported_for(const VptqLayerDesc* desc, const VptqSlicedLayout* layout, int tokens, int flags);
/* 0, or in how many WINDOW PARTS vptq_quant_gemv_sliced_tokens(..., flags) takes these tokens in ONE PASS of the one-token kernel (ABI >= 9,
 * flags | VPTQ_GEMV_EXACT, 2 / 3 tokens): 1 = every column's operands fit beside the slice (no `wstart` needed); 2 / 4 = v = 8 one-table layers
 * whose slice leaves room for half / a quarter of the columns (4096 columns beside a 128 KiB slice, 14336 beside 64 KiB): that many workgroups
 * per (slice, row block), each staging its column windows alone and walking their part of every list (`wstart` needed) */
VPTQ_API int vptq_quant_gemv_sliced_tokens_one_pass(const VptqLayerDesc* desc, int tokens, int flags);
VPTQ_API size_t vptq_quant_gemv_sliced_tokens_workspace_bytes(const VptqLayerDesc* desc, int tokens);
VPTQ_API int vptq_quant_gemv_sliced_tokens(const VptqLayerDesc* desc, const VptqSlicedLayout* layout, const void* x, void* y,
                                  int tokens, int flags, void* workspace, size_t workspace_bytes, void* stream);

/* ... and for up to 3 sibling layers (one format, dtype and input width, the SAME x [tokens][in_features]) in one launch, as
 * vptq_quant_gemv_sliced_grouped does for one token: y / workspaces / workspace_bytes one per layer
 * (vptq_quant_gemv_sliced_tokens_workspace_bytes each) */
VPTQ_API int vptq_quant_gemv_sliced_tokens_grouped(const VptqLayerDesc* descs, const VptqSlicedLayout* layouts, int n, const void* x,
                                          void* const* y, int tokens, int flags, void* const* workspaces,
                                          const size_t* workspace_bytes, void* stream);

/* Up to 3 layers of ONE format, dtype and input width that read the SAME activation (q / k / v, gate / up) in one launch
 * (ABI >= 7): layouts = the layers' structs one after the other (vptq_sliced_layout_tables() each; give every struct the
 * group's rows_per_wave - one round of workgroups over ALL layers), y / workspaces / workspace_bytes one per layer.  The
 * fixed part of a sliced launch (boundary, slice copy, staging, cross-slice hand-over: ~7 of the 10 us of a 4096 x 4096
 * layer) is paid once.
 * flags | VPTQ_GEMV_EXACT | VPTQ_GEMV_COLUMN_PARTS (ABI >= 9): the n descriptors are the COLUMN RANGES [i G / n, (i + 1) G / n) of ONE
 * layer that is too wide for the reference's roundings in one piece (6 bytes of LDS per column: 28672-column layers) - copies of
 * the layer's descriptor with in_features = group_size = G / n (a multiple of 8) and weight_scale / weight_bias / perm /
 * scale_permuted / bias_permuted advanced to the part's first column, a layout per part built from those columns of the index
 * matrix (columns counted from the part's first); x = the WHOLE activation, y[i] = y[0], workspaces[i] = workspaces[0]: the parts
 * meet in the output's accumulator word (n x slices arrivals; at most 127).  vptq_quant_gemv_sliced_tokens_grouped takes the same flags for
 * 2 / 3 tokens where every part takes them in one pass (vptq_quant_gemv_sliced_tokens_one_pass(part, tokens, flags) != 0): x [tokens][G]. */
VPTQ_API int vptq_quant_gemv_sliced_grouped(const VptqLayerDesc* descs, const VptqSlicedLayout* layouts, int n,
                                   const void* x, void* const* y, int flags, void* const* workspaces,
                                   const size_t* workspace_bytes, void* stream);

/* W[O, I] dense, row-major, desc->dtype: the reference CPU path's bits. */
VPTQ_API int vptq_dequant(const VptqLayerDesc* desc, void* W, void* stream);

VPTQ_API int vptq_quant_gemv_v2(const VptqV2Desc* desc, const void* x, void* y, int tokens,
                       int flags, void* stream);

/* name of the kernel vptq_quant_gemv would launch for (desc, tokens, flags);
 * static string, for tests / profiles.  NULL if unsupported. */
VPTQ_API const char* vptq_quant_gemv_kernel_name(const VptqLayerDesc* desc, int tokens, int flags);
/* same for vptq_quant_gemv_grouped (the kernel choice depends on the whole group); "per-layer"
 * when the group is not served by one launch */
VPTQ_API const char* vptq_quant_gemv_grouped_kernel_name(const VptqLayerDesc* descs, int n, int tokens,
                                                int flags);

#ifdef __cplusplus
}
#endif
#endif /* VPTQ_HIP_H */
