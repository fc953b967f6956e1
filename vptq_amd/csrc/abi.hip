// C ABI of libvptq_hip.so (declared in include/vptq_hip.h): argument validation,
// kernel selection, launch.  No allocation, no synchronisation, no torch.
#include <vector>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include "kernels.h"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

int hip_fail(hipError_t e, const char* what) {
  snprintf(g_err, sizeof(g_err), "%s: %s (hipError %d)", what, hipGetErrorString(e), (int)e);
  return (int)e;
}

bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }
int ilog2(int v) { int b = 0; while ((1 << b) < v) ++b; return b; }

// Shape / pointer checks shared by gemv and dequant.  Mirrors what the reference
// asserts with TORCH_CHECK (csrc/quant_gemv.cu:252-282, csrc/dequant.cu:239-275)
// plus the shape algebra of VQuantLinear.__init__ (vptq/layers/vqlinear.py:97-240).
int validate_layer(const VptqLayerDesc* d) {
  if (!d) return fail(VPTQ_E_NULL, "desc is NULL");
  if (!d->indices || !d->centroids) return fail(VPTQ_E_NULL, "indices/centroids is NULL");
  if (d->dtype != VPTQ_DTYPE_F16 && d->dtype != VPTQ_DTYPE_BF16)
    return fail(VPTQ_E_UNSUPPORTED, "dtype %d: only f16 (0) / bf16 (1)", d->dtype);
  const int v = d->vector_len;
  if (!(v == 2 || v == 4 || v == 6 || v == 8 || v == 10 || v == 12 || v == 16))
    return fail(VPTQ_E_UNSUPPORTED, "un-supported vector_len %d", v);
  if (d->in_features <= 0 || d->out_features <= 0 || d->num_codebooks <= 0 || d->group_size <= 0)
    return fail(VPTQ_E_SHAPE, "non-positive dimension");
  if (d->outlier_size < 0 ||
      d->in_features != d->outlier_size + d->num_codebooks * d->group_size)
    return fail(VPTQ_E_SHAPE, "in_features %d != outlier_size %d + %d*%d", d->in_features,
                d->outlier_size, d->num_codebooks, d->group_size);
  if (!pow2(d->num_centroids) || d->num_centroids > 65536)
    return fail(VPTQ_E_SHAPE, "num_centroids %d must be a power of two <= 65536",
                d->num_centroids);
  if (d->index_bits != ilog2(d->num_centroids))
    return fail(VPTQ_E_SHAPE, "index_bits %d != log2(num_centroids)", d->index_bits);
  if (d->num_res_centroids < 0 ||
      (d->num_res_centroids > 0 &&
       (!pow2(d->num_res_centroids) || d->num_res_centroids > 65536)))
    return fail(VPTQ_E_SHAPE, "num_res_centroids %d must be 0 or a power of two <= 65536",
                d->num_res_centroids);
  if (d->res_bits != (d->num_res_centroids > 0 ? ilog2(d->num_res_centroids) : 0))
    return fail(VPTQ_E_SHAPE, "res_bits %d != log2(num_res_centroids)", d->res_bits);
  if ((d->num_res_centroids > 0) != (d->res_centroids != nullptr))
    return fail(VPTQ_E_NULL, "res_centroids must be set iff num_res_centroids > 0");
  const int T = d->index_bits + d->res_bits;
  if (T < 1 || T > 32) return fail(VPTQ_E_SHAPE, "total index bits %d outside [1, 32]", T);
  const long long need_words = ((long long)d->group_size * T + 31) / 32;
  if (d->row_words < need_words)
    return fail(VPTQ_E_SHAPE, "row_words %d < ceil(group_size*bits/32) = %lld", d->row_words,
                need_words);
  if (d->num_indices != (d->out_features + v - 1) / v)
    return fail(VPTQ_E_SHAPE, "num_indices %d != ceil(out_features/vector_len)", d->num_indices);
  // the kernels address the packed indices with 32-bit byte offsets
  if ((unsigned long long)d->num_codebooks * (unsigned long long)d->num_indices * (unsigned long long)d->row_words * 4ull >=
      (1ull << 32))
    return fail(VPTQ_E_SHAPE, "packed indices of %d x %d x %d words: 4 GiB or more are not supported", d->num_codebooks,
                d->num_indices, d->row_words);
  if (d->outlier_size > 0) {
    if (!d->outlier_indices || !d->outlier_centroids)
      return fail(VPTQ_E_NULL, "outlier tensors required when outlier_size > 0");
    if (d->outlier_vector_len < 1 || d->num_outlier_centroids < 1 ||
        d->num_outlier_centroids > 65536)
      return fail(VPTQ_E_SHAPE, "bad outlier codebook shape");
    if (d->num_outlier_indices !=
        (d->out_features + d->outlier_vector_len - 1) / d->outlier_vector_len)
      return fail(VPTQ_E_SHAPE, "num_outlier_indices != ceil(out_features/outlier_vector_len)");
  }
  if ((d->weight_scale != nullptr) != (d->weight_bias != nullptr))
    return fail(VPTQ_E_NULL, "weight_scale and weight_bias must both be set or both NULL");
  if (d->perm && d->in_features > 65536)
    return fail(VPTQ_E_SHAPE, "perm is uint16: in_features must be <= 65536");
  if (((uintptr_t)d->indices | (uintptr_t)d->centroids | (uintptr_t)d->res_centroids) & 3)
    return fail(VPTQ_E_ALIGN, "indices / codebooks must be 4-byte aligned");
  return VPTQ_OK;
}

// VPTQ_GEMV_SELECTIVE where a kernel does not implement it = VPTQ_GEMV_EXACT (always at least as close to the reference);
// EXACT wins when both are set
int selective_as_exact(int flags) {
  return (flags & VPTQ_GEMV_SELECTIVE) ? ((flags & ~VPTQ_GEMV_SELECTIVE) | VPTQ_GEMV_EXACT) : flags;
}
int drop_redundant_selective(int flags) {
  return (flags & VPTQ_GEMV_EXACT) ? (flags & ~VPTQ_GEMV_SELECTIVE) : flags;
}

}  // namespace

extern "C" {

int vptq_abi_version(void) { return VPTQ_ABI_VERSION; }

const char* vptq_last_error(void) { return g_err; }

int vptq_quant_gemv_max_tokens(const VptqLayerDesc* d) {
  if (validate_layer(d) != VPTQ_OK) return 0;
  if (vptq::gemm_k256_eligible(*d, 16, 0)) return 48;
  // bf16: 15.6 us per launch of 16 tokens (gemm_k256t) against ~55 us for dequant + GEMM at 8192^2
  if (vptq::gemm_k256t_eligible(*d, 16, 0)) return 48;
  return vptq::gemv_k256_eligible(*d, 4) ? VPTQ_GEMV_MAX_TOKENS_ANY : 8;
}

// ---- which kernel family serves (layer, tokens, flags): ONE decision for vptq_quant_gemv and for
// vptq_quant_gemv_kernel_name (x = NULL there: the activation pointer is assumed aligned) ----
enum Route { kRouteNone, kRouteGemmK256T, kRouteGemmK256, kRouteK256, kRouteGather, kRouteLds, kRouteGatherX, kRouteGeneric };

static int batch_min_tokens() {
  static std::atomic<int> v{-1};  // VPTQ_GEMM_MIN_TOKENS: smallest token count that takes the batched-decode kernel
  // (at most 16: vptq_quant_gemv_max_tokens promises the batched kernel for 17+ tokens of such layers)
  if (v < 0) { const char* ev = vptq::tune_env("VPTQ_GEMM_MIN_TOKENS"); const int w = ev ? atoi(ev) : 5; v = w > 16 ? 16 : (w < 1 ? 1 : w); }
  return v;
}

// Smallest token count that takes the one-pass batched-decode kernel (gemm_k256t.hip).  Measured at 8192^2 / 4096^2
// (profiles/r03/tokens_*): 13.5 / 9.3 us at 5 tokens ... 15.6 / 10.3 us at 16 for either dtype (its pre-pass is 4.8 us
// of that) against 9.0-10.3 / 5.2-6.6 us for 2-4 tokens on the GEMV kernels, 17.8-18.5 / 10.4-10.8 us for 5-16 fp16
// tokens on gemm_k256 and 18-39 us for bf16 as launches of <= 4 tokens: the default route from 5 tokens for both
// dtypes.  VPTQ_GEMMT_MIN_TOKENS_F16 / _BF16 override (tuning), VPTQ_GEMV_FORCE_BATCHED forces.
static int batch_t_min_tokens(int dtype) {
  static std::atomic<int> vf{-1}, vb{-1};
  std::atomic<int>& v = dtype == VPTQ_DTYPE_F16 ? vf : vb;
  if (v < 0) {
    const char* ev = vptq::tune_env(dtype == VPTQ_DTYPE_F16 ? "VPTQ_GEMMT_MIN_TOKENS_F16" : "VPTQ_GEMMT_MIN_TOKENS_BF16");
    const int w = ev ? atoi(ev) : 5;
    v = w < 1 ? 1 : w;
  }
  return v;
}

// have_ws: the caller's workspace holds vptq_quant_gemv_workspace_bytes (kernel-name queries: assumed)
static Route route_gemv(const VptqLayerDesc& d, int tokens, int flags, const void* x, bool have_ws) {
  const uintptr_t xa = (uintptr_t)x;   // 0 when unknown
  const bool forced_generic = (flags & VPTQ_GEMV_FORCE_GENERIC) != 0;
  // canonical format, folded arithmetic: up to 16 tokens in ONE pass over the indices (launches of 16)
  if (have_ws && !(flags & (VPTQ_GEMV_FORCE_GENERIC | VPTQ_GEMV_FORCE_VALU | VPTQ_GEMV_FORCE_MFMA)) &&
      ((flags & VPTQ_GEMV_FORCE_BATCHED) || tokens >= batch_t_min_tokens(d.dtype)) &&
      vptq::gemm_k256t_eligible(d, tokens > 16 ? 16 : tokens, flags) && (xa & 1) == 0)
    return kRouteGemmK256T;
  // canonical format, fp16, 5-16 tokens: ONE launch of the batched-decode kernel per 16 tokens
  if (!(flags & (VPTQ_GEMV_FORCE_GENERIC | VPTQ_GEMV_FORCE_VALU | VPTQ_GEMV_FORCE_MFMA)) &&
      tokens >= batch_min_tokens() && vptq::gemm_k256_eligible(d, tokens > 16 ? 16 : tokens, flags) && (xa & 15) == 0)
    return kRouteGemmK256;
  if (tokens > VPTQ_GEMV_MAX_TOKENS_ANY) return kRouteNone;
  const int chunk4 = tokens > 4 ? 4 : tokens;
  if (!forced_generic && vptq::gemv_k256_eligible(d, chunk4) && (xa & 15) == 0 &&
      (tokens <= 4 || (d.in_features % 8) == 0))
    return kRouteK256;
  if (!forced_generic && vptq::gemv_gather_eligible(d, tokens > 8 ? 8 : tokens) && (xa & 3) == 0) return kRouteGather;
  if (!forced_generic && vptq::gemv_lds_eligible(d, chunk4, flags) && (xa & 15) == 0) return kRouteLds;
  if (!forced_generic && vptq::gemv_gatherx_eligible(d, chunk4) && (xa & 3) == 0) return kRouteGatherX;   // (4 fits every launch size)
  return kRouteGeneric;
}

size_t vptq_quant_gemv_workspace_bytes(const VptqLayerDesc* d, int tokens, int flags) {
  if (validate_layer(d) != VPTQ_OK || tokens < 1 || tokens > VPTQ_GEMV_MAX_TOKENS) return 0;
  flags = selective_as_exact(flags);
  return route_gemv(*d, tokens, flags, nullptr, true) == kRouteGemmK256T ? vptq::gemm_k256t_workspace_bytes(*d) : 0;
}

const char* vptq_quant_gemv_kernel_name(const VptqLayerDesc* d, int tokens, int flags) {
  if (validate_layer(d) != VPTQ_OK || tokens < 1 || tokens > VPTQ_GEMV_MAX_TOKENS) return nullptr;
  const int kflags = drop_redundant_selective(flags);
  flags = selective_as_exact(flags);
  switch (route_gemv(*d, tokens, flags, nullptr, true)) {
    case kRouteGemmK256T: return "gemm_k256t_kernel";
    case kRouteGemmK256: return "gemm_k256_kernel";
    case kRouteK256: return vptq::gemv_k256_name(*d, tokens, tokens == 1 ? kflags : flags);
    case kRouteGather: return "gemv_gather_kernel";
    case kRouteLds: return vptq::gemv_lds_name(*d, tokens, flags);
    case kRouteGatherX: return "gemv_gatherx_kernel";
    case kRouteGeneric: return "gemv_generic_kernel";
    default: return nullptr;
  }
}

const char* vptq_quant_gemv_grouped_kernel_name(const VptqLayerDesc* descs, int n, int tokens,
                                                int flags) {
  if (!descs || n < 1 || n > VPTQ_GROUP_MAX || tokens < 1 || tokens > VPTQ_GEMV_MAX_TOKENS_ANY) return nullptr;
  bool one_launch = !(flags & VPTQ_GEMV_FORCE_GENERIC);
  for (int i = 0; i < n; ++i) {
    if (validate_layer(&descs[i]) != VPTQ_OK) return nullptr;
    one_launch = one_launch && descs[i].dtype == descs[0].dtype &&
                 vptq::gemv_k256_eligible(descs[i], tokens) &&
                 ((descs[i].perm != nullptr) == (descs[0].perm != nullptr));
  }
  return one_launch ? vptq::gemv_k256_group_name(descs, n, tokens, flags) : "per-layer";
}

int vptq_quant_gemv(const VptqLayerDesc* d, const void* x, void* y, int tokens, int flags,
                    void* workspace, size_t workspace_bytes, void* stream) {
  int rc = validate_layer(d);
  if (rc) return rc;
  // SELECTIVE: the canonical format's kernels decide (gemv_k256.hip:choose_kernel - the persistent MFMA kernel at one token, else
  // the reference's roundings); every other route takes the reference's roundings
  const int kflags = drop_redundant_selective(flags);
  flags = selective_as_exact(flags);
  if (!x || !y) return fail(VPTQ_E_NULL, "x / y is NULL");
  // (a workspace is never required: without it the call takes the kernels that need none)
  const bool have_ws = workspace && (((uintptr_t)workspace) & 15) == 0 &&
                       workspace_bytes >= vptq::gemm_k256t_workspace_bytes(*d);
  if (tokens < 1 || tokens > VPTQ_GEMV_MAX_TOKENS)
    return fail(VPTQ_E_TOKENS, "tokens %d outside [1, %d]: use vptq_dequant + GEMM", tokens,
                VPTQ_GEMV_MAX_TOKENS);
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipSuccess;
  const bool out_f32 = (flags & VPTQ_GEMV_OUT_F32) != 0;
  const size_t yes = out_f32 ? 4 : 2;  // bytes per output element
  const size_t xrow = (size_t)d->in_features * 2, yrow = (size_t)d->out_features * yes;
  // tokens are served in launches of `step` tokens (more tokens = more launches; up to 16 tokens still
  // cheaper than a dense dequant + GEMM for the canonical format, tools/tokens_crossover.py)
  auto chunks = [&](int step, const char* what, auto&& launch) -> int {
    for (int t0 = 0; t0 < tokens; t0 += step) {
      const int m = tokens - t0 < step ? tokens - t0 : step;
      e = launch((const char*)x + (size_t)t0 * xrow, (char*)y + (size_t)t0 * yrow, m);
      if (e != hipSuccess) return hip_fail(e, what);
    }
    return VPTQ_OK;
  };
  switch (route_gemv(*d, tokens, flags, x, have_ws)) {
    case kRouteGemmK256T:  // tokens = the M dimension of a 16x16x32 MFMA fed by transposing gathers; folded arithmetic.
      // (launches of one call share the workspace: they are ordered on the stream)
      return chunks(16, "gemm_k256t launch", [&](const void* xc, void* yc, int m) {
        return vptq::launch_gemm_k256t(*d, xc, yc, m, out_f32, workspace, st); });
    case kRouteGemmK256:   // tokens = the M dimension of a 16x16x16 MFMA; reference arithmetic
      return chunks(16, "gemm_k256 launch", [&](const void* xc, void* yc, int m) {
        return vptq::launch_gemm_k256(*d, xc, yc, m, out_f32, st); });
    case kRouteK256:
      return chunks(4, "gemv_k256 launch", [&](const void* xc, void* yc, int m) {
        return vptq::launch_gemv_k256(d, 1, &xc, &yc, m, tokens == 1 ? kflags : flags, st); });
    case kRouteGather:     // up to 8 tokens per pass over the indices
      return chunks(8, "gemv_gather launch", [&](const void* xc, void* yc, int m) {
        return vptq::launch_gemv_gather(*d, xc, yc, m, out_f32, st); });
    case kRouteLds: {
      // every token of one call in the same arithmetic: the one-token MFMA form only for one-token calls
      const int lflags = tokens > 1 ? (flags | VPTQ_GEMV_EXACT) : flags;
      return chunks(vptq::gemv_lds_max_chunk(d->dtype), "gemv_lds launch", [&](const void* xc, void* yc, int m) {
        return vptq::launch_gemv_lds(*d, xc, yc, m, out_f32, lflags, st); });
    }
    case kRouteGatherX:    // 8 token slots for v <= 8, else 4
      return chunks(vptq::gemv_gatherx_max_chunk(*d), "gemv_gatherx launch", [&](const void* xc, void* yc, int m) {
        return vptq::launch_gemv_gatherx(*d, xc, yc, m, out_f32, st); });
    case kRouteGeneric:    // the generic kernel takes up to 8 tokens per launch
      return chunks(8, "gemv_generic launch", [&](const void* xc, void* yc, int m) {
        return vptq::launch_gemv_generic(*d, xc, yc, m, out_f32, st); });
    default:
      return fail(VPTQ_E_TOKENS, "tokens %d outside [1, %d] for this layer: use vptq_dequant + GEMM", tokens,
                  VPTQ_GEMV_MAX_TOKENS_ANY);
  }
}

int vptq_quant_gemv_grouped(const VptqLayerDesc* descs, int n, const void* const* x,
                            void* const* y, int tokens, int flags, void* stream) {
  if (!descs || !x || !y) return fail(VPTQ_E_NULL, "descs / x / y is NULL");
  if (n < 1 || n > VPTQ_GROUP_MAX) return fail(VPTQ_E_SHAPE, "n %d outside [1, %d]", n, VPTQ_GROUP_MAX);
  if (tokens < 1 || tokens > VPTQ_GEMV_MAX_TOKENS_ANY)
    return fail(VPTQ_E_TOKENS, "tokens %d outside [1, %d]", tokens, VPTQ_GEMV_MAX_TOKENS_ANY);
  flags = tokens == 1 ? drop_redundant_selective(flags) : selective_as_exact(flags);   // (one token: gemv_k256.hip:choose_kernel decides)
  bool all_fast = !(flags & VPTQ_GEMV_FORCE_GENERIC);
  bool same_perm = true;  // one instantiation serves the whole group
  for (int i = 0; i < n; ++i) {
    int rc = validate_layer(&descs[i]);
    if (rc) return rc;
    if (!x[i] || !y[i]) return fail(VPTQ_E_NULL, "x[%d] / y[%d] is NULL", i, i);
    if (descs[i].dtype != descs[0].dtype)
      return fail(VPTQ_E_UNSUPPORTED, "grouped layers must share one dtype");
    all_fast = all_fast && vptq::gemv_k256_eligible(descs[i], tokens) &&
               (((uintptr_t)x[i]) & 15) == 0;
    same_perm = same_perm && ((descs[i].perm != nullptr) == (descs[0].perm != nullptr));
  }
  hipStream_t st = (hipStream_t)stream;
  if (all_fast && !same_perm) {
    for (int i = 0; i < n; ++i) {
      hipError_t e = vptq::launch_gemv_k256(descs + i, 1, x + i, y + i, tokens, flags, st);
      if (e != hipSuccess) return hip_fail(e, "gemv_k256 launch");
    }
    return VPTQ_OK;
  }
  if (all_fast) {
    // one launch for up to 32 layers
    for (int i0 = 0; i0 < n; i0 += 32) {
      const int m = n - i0 < 32 ? n - i0 : 32;
      hipError_t e = vptq::launch_gemv_k256(descs + i0, m, x + i0, y + i0, tokens, flags, st);
      if (e != hipSuccess) return hip_fail(e, "gemv_k256 grouped launch");
    }
    return VPTQ_OK;
  }
  for (int i = 0; i < n; ++i) {
    const int rc = vptq_quant_gemv(&descs[i], x[i], y[i], tokens, flags, nullptr, 0, stream);
    if (rc) return rc;
  }
  return VPTQ_OK;
}

// ---- chain: one persistent launch per <= 32 layers (gemv_k256c.hip), else layer by layer ----
static bool chain_one_kernel(const VptqLayerDesc* descs, int n, const void* const* x, int tokens, int flags) {
  if (tokens != 1 || (flags & (VPTQ_GEMV_FORCE_GENERIC | VPTQ_GEMV_FORCE_VALU))) return false;
  const bool exact = (flags & VPTQ_GEMV_EXACT) != 0, dep = (flags & VPTQ_GEMV_CHAIN_DEPENDENT) != 0;
  const bool sel = !exact && (flags & VPTQ_GEMV_SELECTIVE) != 0;
  for (int i = 0; i < n; ++i) {
    if (descs[i].dtype != descs[0].dtype || !vptq::gemv_k256c_eligible(descs[i], tokens)) return false;
    if (exact && !vptq::gemv_k256c_exact_ok(descs[i], dep)) return false;   // (bf16 / dependent: layer by layer)
    if (sel && !vptq::gemv_k256c_selective_ok(descs[i], dep)) return false;
    if (x && (((uintptr_t)x[i]) & 3) != 0) return false;
  }
  // a chain that cannot keep the workgroups busy (one small layer, q / k / v of a small model) is better
  // served by the per-layer kernels; VPTQ_GEMV_FORCE_MFMA takes the chain kernel regardless (tests)
  if (!(flags & VPTQ_GEMV_FORCE_MFMA)) {
    const bool dependent = (flags & VPTQ_GEMV_CHAIN_DEPENDENT) != 0;
    for (int i0 = 0; i0 < n; i0 += 32)
      if (!vptq::gemv_k256c_fills_device(descs + i0, n - i0 < 32 ? n - i0 : 32, dependent)) return false;
  }
  return true;
}

// SELECTIVE (independent lists): per launch of <= 32 layers a header + one float per output (gemv_k256c.hip), in front of
// everything else
static size_t chain_sel_bytes(const VptqLayerDesc* descs, int n, int flags) {
  flags = drop_redundant_selective(flags);
  if (!(flags & VPTQ_GEMV_SELECTIVE) || (flags & VPTQ_GEMV_CHAIN_DEPENDENT) || !descs || n < 1) return 0;
  size_t b = 0;
  for (int i0 = 0; i0 < n; i0 += 32) b += vptq::gemv_k256c_selective_bytes(descs + i0, n - i0 < 32 ? n - i0 : 32);
  return b;
}
size_t vptq_quant_gemv_chain_workspace_bytes(int n, int flags) {
  return (flags & VPTQ_GEMV_CHAIN_DEPENDENT) && n > 0 ? (size_t)n * 1024 : 0;   // 256 arrival flags per layer
}
// ... and, for an INDEPENDENT list, room for x[perm] of every layer that has an input permutation (optional: without it
// such a list is served by grouped / single launches, which take permutations themselves)
size_t vptq_quant_gemv_chain_workspace_bytes_for(const VptqLayerDesc* descs, int n, int flags) {
  size_t b = vptq_quant_gemv_chain_workspace_bytes(n, flags);
  if (!descs || n < 1 || (flags & VPTQ_GEMV_CHAIN_DEPENDENT)) return b;
  b += chain_sel_bytes(descs, n, flags);
  for (int i = 0; i < n; ++i) b += vptq::gemv_k256c_perm_bytes(descs[i]);
  return b;
}

// How a chain call is executed.  The persistent launch pays its prologue (first codebook image, queue fill: ~7 us)
// once per call and wins from ~16 independent layers on; up to 8 it loses to ONE grouped launch of the one-layer
// kernels (layer = blockIdx.y; 8192^2, us per layer at 2 / 4 / 8 layers: 7.72 / 6.06 / 5.31 against 5.86 / 5.45 / 5.01,
// profiles/r03/chain_vs_grouped_by_length.txt), and a list that cannot fill the device with the persistent kernel
// (q / k / v of a small model) is still better served by one grouped launch than by one launch per layer.
enum ChainRoute { kChainPersistent, kChainGrouped, kChainPerLayer };
constexpr int kChainGroupedMax = 8;
static ChainRoute chain_route(const VptqLayerDesc* descs, int n, const void* const* x, int tokens, int flags) {
  const bool dependent = (flags & VPTQ_GEMV_CHAIN_DEPENDENT) != 0;
  const bool persistent_ok = chain_one_kernel(descs, n, x, tokens, flags);
  if (n == 1 && !(flags & VPTQ_GEMV_FORCE_MFMA)) return kChainPerLayer;   // (the one-layer kernels: arguments preloaded, short prologue)
  if (persistent_ok && (dependent || n > kChainGroupedMax || (flags & VPTQ_GEMV_FORCE_MFMA))) return kChainPersistent;
  if (!dependent && n >= 2 && n <= VPTQ_GROUP_MAX && tokens <= VPTQ_GEMV_MAX_TOKENS_ANY &&
      (n <= kChainGroupedMax || !persistent_ok)) {
    for (int i = 1; i < n; ++i)
      if (descs[i].dtype != descs[0].dtype) return kChainPerLayer;
    return kChainGrouped;
  }
  return persistent_ok ? kChainPersistent : kChainPerLayer;
}

const char* vptq_quant_gemv_chain_kernel_name(const VptqLayerDesc* descs, int n, int tokens, int flags) {
  if (!descs || n < 1 || n > VPTQ_CHAIN_MAX || tokens < 1 || tokens > VPTQ_GEMV_MAX_TOKENS) return nullptr;
  for (int i = 0; i < n; ++i)
    if (validate_layer(&descs[i]) != VPTQ_OK) return nullptr;
  if (!(flags & VPTQ_GEMV_CHAIN_DEPENDENT) && tokens == 1) {
    // (as vptq_quant_gemv_kernel_name: assuming the caller hands over the workspace ..._workspace_bytes_for asks for -
    // layers with an input permutation then stay in the persistent launch)
    bool any = false, ok = true;
    std::vector<VptqLayerDesc> dd(descs, descs + n);
    for (int i = 0; i < n; ++i) {
      if (!dd[i].perm) continue;
      any = true;
      ok = ok && dd[i].scale_permuted && dd[i].bias_permuted && (dd[i].in_features % 8) == 0 && (((uintptr_t)dd[i].perm) & 15) == 0;
      dd[i].weight_scale = dd[i].scale_permuted;
      dd[i].weight_bias = dd[i].bias_permuted;
      dd[i].perm = nullptr; dd[i].inv_perm = nullptr; dd[i].scale_permuted = nullptr; dd[i].bias_permuted = nullptr;
    }
    if (any && ok && chain_route(dd.data(), n, nullptr, tokens, flags) == kChainPersistent) return "gemv_k256c_kernel";
  }
  switch (chain_route(descs, n, nullptr, tokens, flags)) {
    case kChainPersistent: return "gemv_k256c_kernel";
    case kChainGrouped: return "grouped";
    default: return "per-layer";
  }
}

int vptq_quant_gemv_chain(const VptqLayerDesc* descs, int n, const void* const* x, void* const* y,
                          int tokens, int flags, void* workspace, size_t workspace_bytes, void* stream) {
  if (!descs || !x || !y) return fail(VPTQ_E_NULL, "descs / x / y is NULL");
  if (n < 1 || n > VPTQ_CHAIN_MAX) return fail(VPTQ_E_SHAPE, "n %d outside [1, %d]", n, VPTQ_CHAIN_MAX);
  if (tokens < 1 || tokens > VPTQ_GEMV_MAX_TOKENS)
    return fail(VPTQ_E_TOKENS, "tokens %d outside [1, %d]", tokens, VPTQ_GEMV_MAX_TOKENS);
  for (int i = 0; i < n; ++i) {
    const int rc = validate_layer(&descs[i]);
    if (rc) return rc;
    if (!x[i] || !y[i]) return fail(VPTQ_E_NULL, "x[%d] / y[%d] is NULL", i, i);
  }
  const bool dependent = (flags & VPTQ_GEMV_CHAIN_DEPENDENT) != 0;
  flags = drop_redundant_selective(flags);
  // SELECTIVE: the thresholds live in the first chain_thr_bytes of the workspace; without it (or for a dependent list) the
  // call takes the reference's roundings everywhere
  char* thr_ws = nullptr;
  {
    const size_t tb = chain_sel_bytes(descs, n, flags);
    if (flags & VPTQ_GEMV_SELECTIVE) {
      if (tb > 0 && workspace && workspace_bytes >= tb && (((uintptr_t)workspace) & 255) == 0) {
        thr_ws = (char*)workspace;
        workspace = (char*)workspace + tb;
        workspace_bytes -= tb;
      } else {
        flags = selective_as_exact(flags);
      }
    }
  }
  const int lflags = flags & ~VPTQ_GEMV_CHAIN_DEPENDENT;
  hipStream_t st = (hipStream_t)stream;
  // Layers with an input permutation in an independent list: with enough workspace for x[perm] the list still runs in the
  // persistent launch - on (x[perm], scale_permuted, bias_permuted), gathered by one small launch in front of it.
  if (!dependent && tokens == 1) {
    size_t need = 0;
    for (int i = 0; i < n; ++i) need += vptq::gemv_k256c_perm_bytes(descs[i]);
    if (need > 0 && workspace && workspace_bytes >= need && (((uintptr_t)workspace) & 255) == 0) {
      std::vector<VptqLayerDesc> dd(descs, descs + n);
      std::vector<const void*> xx(x, x + n);
      std::vector<void*> xp(n, nullptr);
      char* w = (char*)workspace;
      bool ok = true;
      for (int i = 0; i < n; ++i) {
        if (!dd[i].perm) continue;
        ok = ok && dd[i].scale_permuted && dd[i].bias_permuted && (dd[i].in_features % 8) == 0 && (((uintptr_t)dd[i].perm) & 15) == 0;
        xp[i] = w;
        xx[i] = w;
        w += vptq::gemv_k256c_perm_bytes(dd[i]);
        dd[i].weight_scale = dd[i].scale_permuted;
        dd[i].weight_bias = dd[i].bias_permuted;
        dd[i].perm = nullptr; dd[i].inv_perm = nullptr; dd[i].scale_permuted = nullptr; dd[i].bias_permuted = nullptr;
      }
      if (ok && chain_route(dd.data(), n, xx.data(), tokens, flags) == kChainPersistent) {
        for (int i0 = 0; i0 < n; i0 += 32) {
          const int m = n - i0 < 32 ? n - i0 : 32;
          hipError_t e = vptq::launch_permute_x(descs + i0, m, x + i0, xp.data() + i0, st);
          if (e != hipSuccess) return hip_fail(e, "permute_x launch");
          e = vptq::launch_gemv_k256c(dd.data() + i0, m, xx.data() + i0, y + i0, lflags, false, (uint32_t*)thr_ws, st);
          if (thr_ws) thr_ws += vptq::gemv_k256c_selective_bytes(dd.data() + i0, m);
          if (e != hipSuccess) return hip_fail(e, "gemv_k256c launch");
        }
        return VPTQ_OK;
      }
    }
  }
  const ChainRoute route = chain_route(descs, n, x, tokens, flags);
  if (route == kChainGrouped)   // independent layers, one launch (it serves members it has no kernel for one by one)
    return vptq_quant_gemv_grouped(descs, n, x, y, tokens, lflags & ~VPTQ_GEMV_FORCE_MFMA, stream);
  if (route == kChainPerLayer) {
    // stream order is the dependency
    const int pflags = lflags & ~VPTQ_GEMV_FORCE_MFMA;
    for (int i = 0; i < n; ++i) {
      const int rc = vptq_quant_gemv(&descs[i], x[i], y[i], tokens, pflags, nullptr, 0, stream);
      if (rc) return rc;
    }
    return VPTQ_OK;
  }
  if (dependent) {
    const size_t need = vptq_quant_gemv_chain_workspace_bytes(n, flags);
    if (!workspace || workspace_bytes < need || (((uintptr_t)workspace) & 3) != 0)
      return fail(VPTQ_E_WORKSPACE, "dependent chain: workspace of %zu bytes (4-byte aligned) needed", need);
    const hipError_t e = hipMemsetAsync(workspace, 0, need, st);
    if (e != hipSuccess) return hip_fail(e, "chain workspace clear");
  }
  // VPTQ_K256C_PROF builds write per-wave profile words to the workspace of a non-dependent launch: only when the
  // caller handed over enough of it (256 workgroups x 16 waves x 64 words of 8 bytes)
  const bool prof_ws = !dependent && vptq::tune_env("VPTQ_K256C_PROF") && workspace && workspace_bytes >= (size_t)256 * 16 * 64 * 8;
  for (int i0 = 0; i0 < n; i0 += 32) {
    const int m = n - i0 < 32 ? n - i0 : 32;
    const hipError_t e = vptq::launch_gemv_k256c(descs + i0, m, x + i0, y + i0, lflags, dependent,
                                                 dependent ? (uint32_t*)workspace + (size_t)i0 * 256
                                                           : thr_ws ? (uint32_t*)thr_ws
                                                           : (prof_ws ? (uint32_t*)workspace : nullptr), st);
    if (thr_ws) thr_ws += vptq::gemv_k256c_selective_bytes(descs + i0, m);
    if (dependent && e == hipErrorCooperativeLaunchTooLarge) {
      // the runtime does not confirm that every workgroup of the dependent chain is resident at once (gemv_k256c.hip:launch_c):
      // one launch per layer, stream order is the dependency - slower, never wrong (layers already walked are walked again)
      (void)hipGetLastError();
      const int pflags = lflags & ~VPTQ_GEMV_FORCE_MFMA;
      for (int i = 0; i < n; ++i) {
        const int rc = vptq_quant_gemv(&descs[i], x[i], y[i], tokens, pflags, nullptr, 0, stream);
        if (rc) return rc;
      }
      return VPTQ_OK;
    }
    if (e != hipSuccess) return hip_fail(e, "gemv_k256c launch");
  }
  return VPTQ_OK;
}

int vptq_quant_gemm_supported(const VptqLayerDesc* d) {
  return validate_layer(d) == VPTQ_OK && vptq::gemm_fused_eligible(*d) ? 1 : 0;
}

int vptq_sliced_layout_supported(const VptqLayerDesc* d) {
  return validate_layer(d) == VPTQ_OK && vptq::gemv_sliced_eligible(*d) ? vptq::gemv_sliced_slices(*d) : 0;
}

int vptq_sliced_layout_supported_for(const VptqLayerDesc* d, int flags) {
  const bool exact = (flags & VPTQ_GEMV_EXACT) != 0;
  if (flags & VPTQ_GEMV_FORCE_GENERIC) return 0;
  return validate_layer(d) == VPTQ_OK && vptq::gemv_sliced_eligible(*d, exact) ? vptq::gemv_sliced_slices(*d, exact) : 0;
}

int vptq_sliced_layout_tables(const VptqLayerDesc* d) {
  return validate_layer(d) == VPTQ_OK && vptq::gemv_sliced_eligible(*d) ? vptq::gemv_sliced_tables(*d) : 0;
}

int vptq_sliced_layout_whole_table(const VptqLayerDesc* d, int table) {
  return validate_layer(d) == VPTQ_OK && vptq::gemv_sliced_eligible(*d) ? vptq::gemv_sliced_whole_table(*d, table) : 0;
}

size_t vptq_quant_gemv_sliced_workspace_bytes(const VptqLayerDesc* d) {
  return validate_layer(d) == VPTQ_OK && vptq::gemv_sliced_eligible(*d) ? vptq::gemv_sliced_workspace_bytes(*d) : 0;
}

int vptq_quant_gemv_sliced(const VptqLayerDesc* d, const VptqSlicedLayout* layout, const void* x, void* y, int flags,
                           void* workspace, size_t workspace_bytes, void* stream) {
  int rc = validate_layer(d);
  if (rc) return rc;
  if (!x || !y || !layout) return fail(VPTQ_E_NULL, "x / y / layout is NULL");
  flags = drop_redundant_selective(flags);
  const bool exact = (flags & VPTQ_GEMV_EXACT) != 0;
  const bool sel = (flags & VPTQ_GEMV_SELECTIVE) != 0;
  if (flags & VPTQ_GEMV_FORCE_GENERIC) return fail(VPTQ_E_UNSUPPORTED, "VPTQ_GEMV_FORCE_GENERIC: use vptq_quant_gemv");
  if (sel && !vptq::gemv_hot_eligible(*d))
    return fail(VPTQ_E_UNSUPPORTED, "VPTQ_GEMV_SELECTIVE over a sliced layout: fp16 layers the folded sliced kernel serves, with scale and bias "
                                    "(vptq_quant_gemv_sliced_selective_supported); ask for VPTQ_GEMV_EXACT over an exact layout instead");
  if (!vptq::gemv_sliced_eligible(*d, exact))
    return fail(VPTQ_E_UNSUPPORTED, exact ? "VPTQ_GEMV_EXACT over a sliced layout: v = 8 / 16, 16384 ... 65536 main centroids, no residual "
                                            "codebook or the 256-entry one of v = 8, scale / bias / x of every column beside a slice in LDS "
                                            "(vptq_sliced_layout_supported_for)"
                                          : "the sliced layout serves v = 8 / 16 layers with 16384 ... 65536 main centroids, group_size <= 32768");
  const size_t acc_bytes = (vptq::gemv_sliced_workspace_bytes(*d) + 255) / 256 * 256;
  const size_t need = sel ? acc_bytes + vptq::gemv_hot_bytes(*d) : vptq::gemv_sliced_workspace_bytes(*d);
  if (!workspace || workspace_bytes < need || (((uintptr_t)workspace) & (sel ? 255 : 15)) != 0)
    return fail(VPTQ_E_WORKSPACE, "workspace of %zu bytes (%d-byte aligned) needed", need, sel ? 256 : 16);
  // (folded, two tables: one layout per table, consecutive structs; the reference's roundings: always ONE layout)
  const int n_layouts = exact ? 1 : vptq::gemv_sliced_tables(*d);
  for (int i = 0; i < n_layouts; ++i) {
    if (layout[i].rows_per_wave < 1 || layout[i].rows_per_wave > 64 || !layout[i].elems || !layout[i].blocks || !layout[i].first ||
        (layout[i].n_slices != 0 ? layout[i].n_slices : 8) != vptq::gemv_sliced_slices(*d, exact))
      return fail(VPTQ_E_UNSUPPORTED, "sliced layout %d: rows_per_wave in [1, 64], three tensors, n_slices = %d for this layer and arithmetic",
                  i, vptq::gemv_sliced_slices(*d, exact));
    if (exact && d->num_res_centroids > 0 && (!layout[i].res || (((uintptr_t)layout[i].res) & 1) != 0))
      return fail(VPTQ_E_UNSUPPORTED, "VPTQ_GEMV_EXACT over a sliced layout of a layer with a residual codebook needs the layout's `res` "
                  "side stream (uint8 for v = 8 with 256 residual centroids, else uint16)");
    if (layout[i].whole_table != (exact ? 0 : vptq::gemv_sliced_whole_table(*d, i)) || layout[i].rows_per_wave != layout[0].rows_per_wave)
      return fail(VPTQ_E_UNSUPPORTED, "sliced layout %d: whole_table must be %d (vptq_sliced_layout_whole_table) and rows_per_wave the "
                  "same for both tables", i, vptq::gemv_sliced_whole_table(*d, i));
  }
  if ((((uintptr_t)x) & 15) != 0) return fail(VPTQ_E_UNSUPPORTED, "x must be 16-byte aligned");
  if (sel) {
    // the pre-pass (gemv_hot.hip): threshold, x with the hot blocks' features zeroed, the hot blocks' exact products - behind the
    // accumulator words of the same workspace; then the folded launch over x_masked, which adds the products before its rounding
    const void* xm = nullptr;
    const float* corr = nullptr;
    hipError_t e = vptq::launch_gemv_hot(*d, x, (char*)workspace + acc_bytes, &xm, &corr, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "gemv_hot launch");
    e = vptq::launch_gemv_sliced(*d, layout, xm, y, flags & ~VPTQ_GEMV_SELECTIVE, workspace, (hipStream_t)stream, corr);
    return e == hipSuccess ? VPTQ_OK : hip_fail(e, "gemv_sliced launch");
  }
  const hipError_t e = vptq::launch_gemv_sliced(*d, layout, x, y, flags, workspace, (hipStream_t)stream);
  return e == hipSuccess ? VPTQ_OK : hip_fail(e, "gemv_sliced launch");
}

int vptq_quant_gemv_sliced_selective_supported(const VptqLayerDesc* d) {
  return validate_layer(d) == VPTQ_OK && vptq::gemv_hot_eligible(*d) ? 1 : 0;
}
size_t vptq_quant_gemv_sliced_workspace_bytes_for(const VptqLayerDesc* d, int flags) {
  if (validate_layer(d) != VPTQ_OK || !vptq::gemv_sliced_eligible(*d, (flags & VPTQ_GEMV_EXACT) != 0)) return 0;
  flags = drop_redundant_selective(flags);
  if (!(flags & VPTQ_GEMV_SELECTIVE)) return vptq::gemv_sliced_workspace_bytes(*d);
  return vptq::gemv_hot_eligible(*d) ? (vptq::gemv_sliced_workspace_bytes(*d) + 255) / 256 * 256 + vptq::gemv_hot_bytes(*d) : 0;
}

int vptq_quant_gemv_sliced_tokens_supported(const VptqLayerDesc* d, const VptqSlicedLayout* layout, int tokens) {
  return vptq_quant_gemv_sliced_tokens_supported_for(d, layout, tokens, 0);
}
int vptq_quant_gemv_sliced_tokens_supported_for(const VptqLayerDesc* d, const VptqSlicedLayout* layout, int tokens, int flags) {
  if (flags & ~(VPTQ_GEMV_EXACT | VPTQ_GEMV_OUT_F32)) return 0;
  return validate_layer(d) == VPTQ_OK && layout && vptq::gemv_sliced_tok_eligible(*d, layout, tokens, (flags & VPTQ_GEMV_EXACT) != 0) ? 1 : 0;
}

int vptq_quant_gemv_sliced_tokens_one_pass(const VptqLayerDesc* d, int tokens, int flags) {
  if (validate_layer(d) != VPTQ_OK || (flags & ~(VPTQ_GEMV_EXACT | VPTQ_GEMV_OUT_F32))) return 0;
  return vptq::gemv_sliced_tok_one_pass_parts(*d, tokens, (flags & VPTQ_GEMV_EXACT) != 0);
}

size_t vptq_quant_gemv_sliced_tokens_workspace_bytes(const VptqLayerDesc* d, int tokens) {
  return validate_layer(d) == VPTQ_OK && (vptq::gemv_sliced_eligible(*d) || vptq::gemv_sliced_eligible(*d, true)) && tokens >= 2 && tokens <= 8
             ? vptq::gemv_sliced_tok_workspace_bytes(*d, tokens) : 0;
}

int vptq_quant_gemv_sliced_tokens(const VptqLayerDesc* d, const VptqSlicedLayout* layout, const void* x, void* y, int tokens,
                                  int flags, void* workspace, size_t workspace_bytes, void* stream) {
  if (int rc = validate_layer(d)) return rc;
  if (!layout || !x || !y) return fail(VPTQ_E_NULL, "layout, x and y must be set");
  if (flags & VPTQ_GEMV_FORCE_GENERIC) return fail(VPTQ_E_UNSUPPORTED, "the sliced path is not the generic kernel: use vptq_quant_gemv");
  const bool exact = (flags & VPTQ_GEMV_EXACT) != 0;
  if (!vptq::gemv_sliced_eligible(*d, exact) || !vptq::gemv_sliced_tok_eligible(*d, layout, tokens, exact))
    return fail(VPTQ_E_UNSUPPORTED, "sliced layouts with column windows (wstart), 2 - 8 tokens, and activations that fit the LDS beside the slice"
                                    " (VPTQ_GEMV_EXACT: one-table formats, a layout of vptq_sliced_layout_supported_for(desc, VPTQ_GEMV_EXACT) slices)");
  const size_t need = vptq::gemv_sliced_tok_workspace_bytes(*d, tokens);
  if (!workspace || workspace_bytes < need || (((uintptr_t)workspace) & 15) != 0)
    return fail(VPTQ_E_WORKSPACE, "the sliced path for %d tokens needs %zu bytes of 16-byte aligned, zero-initialised workspace", tokens, need);
  if ((((uintptr_t)x) & 15) != 0) return fail(VPTQ_E_UNSUPPORTED, "x must be 16-byte aligned");
  const hipError_t e = vptq::launch_gemv_sliced_tok(*d, layout, x, y, tokens, flags, workspace, (hipStream_t)stream);
  return e == hipSuccess ? VPTQ_OK : hip_fail(e, "gemv_sliced_tok launch");
}

int vptq_quant_gemv_sliced_grouped(const VptqLayerDesc* descs, const VptqSlicedLayout* layouts, int n, const void* x,
                                   void* const* y, int flags, void* const* workspaces, const size_t* workspace_bytes, void* stream) {
  if (!descs || !layouts || !x || !y || !workspaces || !workspace_bytes) return fail(VPTQ_E_NULL, "descs / layouts / x / y / workspaces is NULL");
  if (n < 1 || n > 3) return fail(VPTQ_E_SHAPE, "n %d outside [1, 3]", n);
  for (int i = 0; i < n; ++i) {
    const int rc = validate_layer(&descs[i]);
    if (rc) return rc;
    if (!y[i]) return fail(VPTQ_E_NULL, "y[%d] is NULL", i);
  }
  const bool exact = (flags & VPTQ_GEMV_EXACT) != 0;
  if (flags & VPTQ_GEMV_FORCE_GENERIC) return fail(VPTQ_E_UNSUPPORTED, "VPTQ_GEMV_FORCE_GENERIC: use vptq_quant_gemv");
  if (flags & VPTQ_GEMV_COLUMN_PARTS) {
    if (!exact) return fail(VPTQ_E_UNSUPPORTED, "VPTQ_GEMV_COLUMN_PARTS goes with VPTQ_GEMV_EXACT (the folded form stages 32768 columns in one piece)");
    for (int i = 1; i < n; ++i)
      if (y[i] != y[0] || workspaces[i] != workspaces[0] || descs[i].out_features != descs[0].out_features ||
          descs[i].num_indices != descs[0].num_indices || descs[i].bias != descs[0].bias || descs[i].in_features != descs[0].in_features)
        return fail(VPTQ_E_UNSUPPORTED, "column parts share y, the workspace, the output bias and have one width");
  }
  if (!vptq::gemv_sliced_groupable(descs, n, exact))
    return fail(VPTQ_E_UNSUPPORTED, "a sliced group takes layers of ONE format, dtype and input width that vptq_sliced_layout_supported_for() accepts");
  if ((((uintptr_t)x) & 15) != 0) return fail(VPTQ_E_UNSUPPORTED, "x must be 16-byte aligned");
  const int tables = exact ? 1 : vptq::gemv_sliced_tables(descs[0]);
  for (int i = 0; i < n; ++i) {
    const size_t need = vptq::gemv_sliced_workspace_bytes(descs[i]);
    if (!workspaces[i] || workspace_bytes[i] < need || (((uintptr_t)workspaces[i]) & 15) != 0)
      return fail(VPTQ_E_WORKSPACE, "layer %d: workspace of %zu bytes (16-byte aligned) needed", i, need);
    for (int t = 0; t < tables; ++t) {
      const VptqSlicedLayout& L = layouts[(size_t)i * tables + t];
      if (L.rows_per_wave < 1 || L.rows_per_wave > 64 || !L.elems || !L.blocks || !L.first ||
          (L.n_slices != 0 ? L.n_slices : 8) != vptq::gemv_sliced_slices(descs[i], exact))
        return fail(VPTQ_E_UNSUPPORTED, "layer %d, sliced layout %d: rows_per_wave in [1, 64], three tensors, n_slices = %d", i, t,
                    vptq::gemv_sliced_slices(descs[i], exact));
      if (exact && descs[i].num_res_centroids > 0 && (!L.res || (((uintptr_t)L.res) & 1) != 0))
        return fail(VPTQ_E_UNSUPPORTED, "layer %d: VPTQ_GEMV_EXACT needs the layout's `res` side stream", i);
      if (L.whole_table != (exact ? 0 : vptq::gemv_sliced_whole_table(descs[i], t)) || L.rows_per_wave != layouts[0].rows_per_wave)
        return fail(VPTQ_E_UNSUPPORTED, "layer %d, sliced layout %d: whole_table must be %d and rows_per_wave the group's", i, t,
                    vptq::gemv_sliced_whole_table(descs[i], t));
    }
  }
  const hipError_t e = vptq::launch_gemv_sliced_group(descs, layouts, n, x, y, flags, workspaces, (hipStream_t)stream);
  return e == hipSuccess ? VPTQ_OK : hip_fail(e, "gemv_sliced grouped launch");
}

int vptq_quant_gemv_sliced_tokens_grouped(const VptqLayerDesc* descs, const VptqSlicedLayout* layouts, int n, const void* x,
                                          void* const* y, int tokens, int flags, void* const* workspaces, const size_t* workspace_bytes,
                                          void* stream) {
  if (!descs || !layouts || !x || !y || !workspaces || !workspace_bytes) return fail(VPTQ_E_NULL, "descs / layouts / x / y / workspaces is NULL");
  if (n < 1 || n > 3) return fail(VPTQ_E_UNSUPPORTED, "a sliced group takes 1 .. 3 layers");
  for (int i = 0; i < n; ++i) {
    if (int rc = validate_layer(descs + i)) return rc;
    if (!y[i]) return fail(VPTQ_E_NULL, "y[%d] is NULL", i);
  }
  if (flags & VPTQ_GEMV_FORCE_GENERIC) return fail(VPTQ_E_UNSUPPORTED, "the sliced path is not the generic kernel: use vptq_quant_gemv");
  if (flags & VPTQ_GEMV_COLUMN_PARTS) {   // (as vptq_quant_gemv_sliced_grouped: parts of ONE layer; 2 / 3 tokens, where every part takes them in one pass)
    if (!(flags & VPTQ_GEMV_EXACT)) return fail(VPTQ_E_UNSUPPORTED, "VPTQ_GEMV_COLUMN_PARTS goes with VPTQ_GEMV_EXACT");
    for (int i = 0; i < n; ++i)
      if (y[i] != y[0] || workspaces[i] != workspaces[0] || descs[i].out_features != descs[0].out_features || descs[i].num_indices != descs[0].num_indices ||
          descs[i].bias != descs[0].bias || descs[i].in_features != descs[0].in_features || !vptq::gemv_sliced_tok_one_pass_parts(descs[i], tokens, true))
        return fail(VPTQ_E_UNSUPPORTED, "column parts share y, the workspace, the output bias, have one width and take the tokens in one pass");
  }
  if (!vptq::gemv_sliced_tok_groupable(descs, layouts, n, tokens, (flags & VPTQ_GEMV_EXACT) != 0))
    return fail(VPTQ_E_UNSUPPORTED, "a sliced group of 2 - 4 tokens takes layers of ONE format, dtype and input width whose layouts carry wstart");
  if ((((uintptr_t)x) & 15) != 0) return fail(VPTQ_E_UNSUPPORTED, "x must be 16-byte aligned");
  for (int i = 0; i < n; ++i) {
    const size_t need = vptq::gemv_sliced_tok_workspace_bytes(descs[i], tokens);
    if (!workspaces[i] || workspace_bytes[i] < need || (((uintptr_t)workspaces[i]) & 15) != 0)
      return fail(VPTQ_E_WORKSPACE, "layer %d: workspace of %zu bytes (16-byte aligned, zero-initialised) needed for %d tokens", i, need, tokens);
  }
  const hipError_t e = vptq::launch_gemv_sliced_tok_group(descs, layouts, n, x, y, tokens, flags, workspaces, (hipStream_t)stream);
  return e == hipSuccess ? VPTQ_OK : hip_fail(e, "gemv_sliced_tok grouped launch");
}

size_t vptq_quant_gemm_workspace_bytes(const VptqLayerDesc* d, int tokens) {
  if (validate_layer(d) != VPTQ_OK || tokens < 1) return 0;
  return vptq::gemm_fused_workspace_bytes(*d, tokens);
}

int vptq_quant_gemm(const VptqLayerDesc* d, const void* x, void* y, int tokens, int flags, void* workspace,
                    size_t workspace_bytes, void* stream) {
  (void)flags;
  int rc = validate_layer(d);
  if (rc) return rc;
  if (!x || !y) return fail(VPTQ_E_NULL, "x / y is NULL");
  if (tokens < 1) return fail(VPTQ_E_TOKENS, "tokens %d < 1", tokens);
  if (!vptq::gemm_fused_eligible(*d))
    return fail(VPTQ_E_UNSUPPORTED, "no fused GEMM for this layer: use vptq_dequant + a dense GEMM");
  if ((((uintptr_t)x) & 15) != 0) return fail(VPTQ_E_ALIGN, "x must be 16-byte aligned");
  if (workspace_bytes < vptq::gemm_fused_workspace_bytes(*d, tokens) ||
      (vptq::gemm_fused_workspace_bytes(*d, tokens) > 0 && !workspace))
    return fail(VPTQ_E_WORKSPACE, "workspace of %zu bytes needed", vptq::gemm_fused_workspace_bytes(*d, tokens));
  hipError_t e = vptq::launch_gemm_fused(*d, x, y, tokens, workspace, workspace_bytes, (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(e, "gemm_fused launch");
  return VPTQ_OK;
}

int vptq_dequant(const VptqLayerDesc* d, void* W, void* stream) {
  int rc = validate_layer(d);
  if (rc) return rc;
  if (!W) return fail(VPTQ_E_NULL, "W is NULL");
  if (d->perm && !d->inv_perm)
    return fail(VPTQ_E_NULL, "dequant needs inv_perm = argsort(perm) when perm is set");
  hipError_t e = vptq::launch_dequant(*d, W, (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(e, "dequant launch");
  return VPTQ_OK;
}

int vptq_quant_gemv_v2(const VptqV2Desc* d, const void* x, void* y, int tokens, int flags,
                       void* stream) {
  const bool out_f32 = (flags & VPTQ_GEMV_OUT_F32) != 0;
  if (!d) return fail(VPTQ_E_NULL, "desc is NULL");
  if (!d->indices || !d->centroids || !x || !y)
    return fail(VPTQ_E_NULL, "indices / centroids / x / y is NULL");
  if (d->dtype != VPTQ_DTYPE_F16 && d->dtype != VPTQ_DTYPE_BF16)
    return fail(VPTQ_E_UNSUPPORTED, "dtype %d", d->dtype);
  // the reference instantiates v in {4, 8, 16} (csrc/dispatch_macros.h:12-89)
  if (!(d->vector_len == 4 || d->vector_len == 8 || d->vector_len == 16))
    return fail(VPTQ_E_UNSUPPORTED, "un-supported vector_len %d (v2: 4, 8, 16)", d->vector_len);
  if (d->in_features <= 0 || d->out_features <= 0 || d->out_features % d->vector_len)
    return fail(VPTQ_E_SHAPE, "out_features must be a positive multiple of vector_len");
  if (d->num_centroids < 1 || d->num_centroids > 65536)
    return fail(VPTQ_E_SHAPE, "num_centroids %d", d->num_centroids);
  if (d->num_res_centroids > 0) {
    if (!d->res_indices || !d->res_centroids)
      return fail(VPTQ_E_NULL, "residual tensors required when num_res_centroids > 0");
    if (d->res_index_bytes != 1 && d->res_index_bytes != 2)
      return fail(VPTQ_E_SHAPE, "res_index_bytes must be 1 or 2");
    if (d->res_index_bytes == 1 && d->num_res_centroids > 256)
      return fail(VPTQ_E_SHAPE, "uint8 residual ids need num_res_centroids <= 256");
  }
  // reference: "tokens < 16" (vptq/ops/quant_gemm.py:338, csrc/quant_gemv_v2.cu:58)
  if (tokens < 1 || tokens >= 16)
    return fail(VPTQ_E_TOKENS, "tokens %d outside [1, 15]", tokens);
  hipStream_t st = (hipStream_t)stream;
  const size_t es = 2;
  if (!(flags & VPTQ_GEMV_FORCE_GENERIC) && vptq::gemv_lds_v2_eligible(*d, tokens > 4 ? 4 : tokens) &&
      (((uintptr_t)x) & 15) == 0) {
    // codebooks LDS-resident (what the reference's kernel does, quant_gemv_v2.cuh:85-94)
    const int step = vptq::gemv_lds_max_chunk(d->dtype);
    const int lflags = tokens > 1 ? (flags | VPTQ_GEMV_EXACT) : flags;
    for (int t0 = 0; t0 < tokens; t0 += step) {
      const int m = tokens - t0 < step ? tokens - t0 : step;
      hipError_t e = vptq::launch_gemv_lds_v2(*d, (const char*)x + (size_t)t0 * d->in_features * es,
                                              (char*)y + (size_t)t0 * d->out_features * (out_f32 ? 4 : es),
                                              m, out_f32, lflags, st);
      if (e != hipSuccess) return hip_fail(e, "gemv_lds (v2) launch");
    }
    return VPTQ_OK;
  }
  for (int t0 = 0; t0 < tokens; t0 += 8) {
    const int m = tokens - t0 < 8 ? tokens - t0 : 8;
    hipError_t e = vptq::launch_gemv_v2(*d, (const char*)x + (size_t)t0 * d->in_features * es,
                                        (char*)y + (size_t)t0 * d->out_features * (out_f32 ? 4 : es), m,
                                        out_f32, st);
    if (e != hipSuccess) return hip_fail(e, "gemv_v2 launch");
  }
  return VPTQ_OK;
}

}  // extern "C"
