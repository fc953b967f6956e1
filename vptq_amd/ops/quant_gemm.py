"""`vptq.ops` operator API on MI355X: same names, arguments and error behaviour as
the reference (vptq/ops/quant_gemm.py:6-10, :43-70, :161-187, :278-291), backed by
hand-written HIP kernels through the C ABI in include/vptq_hip.h.

Differences from the reference, on purpose:
* no torch fallback — a missing library or a CPU tensor raises;
* the fused GEMV serves 1..8 tokens, the canonical 256+256 format 1..16 (`vptq_quant_gemv_max_tokens`;
  the reference: < 3, quant_gemm.py:213)
  because one workgroup reuses each rebuilt weight for every token, the
  kernel stays HBM-bound; above that, dequant + `F.linear` (hipBLASLt);
* `argsort(perm)` is cached instead of recomputed per call (quant_gemm.py:208-211).
"""
from __future__ import annotations

import os
from typing import Optional

import torch
from torch.nn import functional as F

from vptq_amd import _backend as B

__all__ = ["dequant", "quant_gemm", "quant_gemv_v2"]

# env knob: VPTQ_EXACT=1 forces the reference CPU path's three 16-bit roundings per weight whatever the arithmetic mode
# says (`_backend.set_arithmetic`: "reference" is the default since round 5, "folded" the opt-in fast form)
_FLAGS = B.GEMV_EXACT if B.tune_env("VPTQ_EXACT", "0") == "1" else 0

# The arithmetic of the functional op (`_backend.folded_form_is_safe`): the reference's roundings unless the folded form is
# opted in AND the layer passes its measured gate.  Decided once per set of tensor
# OBJECTS (weak references + version counters): one device -> host read, not one per call.
_GATE_CACHE = {}


def _safe_flags(indices, centroids, residual_centroids, weight_scale, weight_bias, desc=None, in_features=0, out_features=0) -> int:
    if weight_scale is None or weight_bias is None or not weight_scale.is_cuda:
        return 0
    import weakref
    tensors = (indices, centroids, residual_centroids, weight_scale, weight_bias)
    key = tuple(id(t) for t in tensors) + (B.arithmetic_generation(),)
    ent = _GATE_CACHE.get(key)
    if ent is not None:   # (the same OBJECTS, alive and unchanged: a recycled id or storage pointer must not hit)
        refs, vers, hit = ent
        if all((r is None and t is None) or (r is not None and r() is t) for r, t in zip(refs, tensors)) and \
                vers == tuple(B.tensor_version(t) if t is not None else 0 for t in tensors):
            return hit
    hit = B.layer_arithmetic_flags(B.folded_form_is_safe(indices, centroids, residual_centroids, weight_scale, weight_bias, desc,
                                                         in_features, out_features))
    if len(_GATE_CACHE) > 4096:
        _GATE_CACHE.clear()
    _GATE_CACHE[key] = (tuple(weakref.ref(t) if t is not None else None for t in tensors),
                        tuple(B.tensor_version(t) if t is not None else 0 for t in tensors), hit)
    return hit
# Most tokens the fused dequant + GEMM kernel (vptq_quant_gemm, gemm_fused.hip) is used for by
# `quant_gemm`; above it: HIP dequant + hipBLASLt.  0 = never, the default: measured on MI355X
# (tools/prefill_bench.py, profiles/r02/prefill_fused_vs_dense.json) the fused kernel ties the dense
# route around 512-1024 tokens and loses elsewhere (0.84 vs 1.39 PFLOP/s at 8192 tokens), so the
# vendor GEMM stays the default; VPTQ_FUSED_GEMM_MAX_TOKENS=<n> routes up to n tokens through it
# (no dense W is materialised: 2 O I bytes less memory per call).
_FUSED_GEMM_MAX_TOKENS = int(os.environ.get("VPTQ_FUSED_GEMM_MAX_TOKENS", "0"))


def quant_gemm_flags() -> int:
    """flags passed to vptq_quant_gemv by this process (VPTQ_EXACT=1 -> reference roundings)"""
    return _FLAGS


def dequant(
    indices: torch.Tensor,
    centroids: torch.Tensor,
    outlier_indices: Optional[torch.Tensor],
    outlier_centroids: Optional[torch.Tensor],
    res_indices: Optional[torch.Tensor],
    res_centroids: Optional[torch.Tensor],
    perm: Optional[torch.Tensor],
    weight_scale: Optional[torch.Tensor],
    weight_bias: Optional[torch.Tensor],
    is_indice_packed: bool,
    enable_outlier: bool,
    enable_residual: bool,
    enable_perm: bool,
    enable_norm: bool,
    num_centroids: int,
    num_outlier_centroids: int,
    num_res_centroids: int,
    padding: int,
    outlier_padding: int,
    num_codebooks: int,
    group_size: int,
    outlier_size: int,
    vector_len: int,
    outlier_vector_len: int,
    vector_quant_dim: str = "out",
) -> torch.Tensor:
    """Dense weight W[out_features, in_features] of a packed VQuantLinear layer.

    Same signature as the reference's torch implementation
    (vptq/ops/quant_gemm.py:43-158); runs `vptq_dequant` on the GPU and returns
    the same bits the reference CPU path produces.
    """
    if vector_quant_dim == "in":
        raise ValueError("Not implemented yet.")
    if not is_indice_packed:
        raise RuntimeError("vptq_amd handles packed int32 indices only "
                           "(is_indice_packed=True, the format HF checkpoints use)")
    if res_indices is not None:
        raise RuntimeError("packed layers carry residual indices inside `indices`")
    num_indices = indices.shape[1]
    out_features = num_indices * vector_len - padding
    in_features = (outlier_size if enable_outlier else 0) + num_codebooks * group_size
    dev = B.require_device(indices, centroids, outlier_indices, outlier_centroids,
                           res_centroids, perm, weight_scale, weight_bias)
    desc, keep = B.make_layer_desc(
        indices=indices, centroids=centroids,
        res_centroids=res_centroids if enable_residual else None,
        outlier_indices=outlier_indices if enable_outlier else None,
        outlier_centroids=outlier_centroids if enable_outlier else None,
        perm=perm if enable_perm else None,
        weight_scale=weight_scale if enable_norm else None,
        weight_bias=weight_bias if enable_norm else None, bias=None,
        in_features=in_features, out_features=out_features, vector_len=vector_len,
        num_codebooks=num_codebooks, num_centroids=num_centroids,
        num_res_centroids=num_res_centroids if enable_residual else 0,
        group_size=group_size, outlier_size=outlier_size if enable_outlier else 0,
        outlier_vector_len=outlier_vector_len, num_outlier_centroids=num_outlier_centroids,
        need_inv_perm=True)
    W = torch.empty((out_features, in_features), dtype=centroids.dtype, device=dev)
    with torch.cuda.device(dev):
        B.check(B.lib().vptq_dequant(desc, W.data_ptr(), B.current_stream_ptr(dev)),
                "vptq_dequant")
    del keep
    return W


def fused_gemm_max_tokens() -> int:
    """token count up to which many-token calls take the fused dequant-GEMM (VPTQ_FUSED_GEMM_MAX_TOKENS; 0 = never)"""
    return _FUSED_GEMM_MAX_TOKENS


def quant_gemm(
    x: torch.Tensor,
    bias: Optional[torch.Tensor],
    indices: torch.Tensor,
    centroids: torch.Tensor,
    outlier_indices: Optional[torch.Tensor],
    outlier_centroids: Optional[torch.Tensor],
    residual_indices: Optional[torch.Tensor],
    residual_centroids: Optional[torch.Tensor],
    perm: Optional[torch.Tensor],
    weight_scale: Optional[torch.Tensor],
    weight_bias: Optional[torch.Tensor],
    vector_len: int,
    outlier_vector_len: int,
    num_codebooks: int,
    num_centroids: int,
    num_outlier_centroids: int,
    num_res_centroids: int,
    is_indice_packed: bool,
    group_size: int,
    outlier_size: int,
    in_features: int,
    out_features: int,
    padding: int,
    outlier_padding: int,
    vector_quant_dim: str = "out",
    prefetch: Optional[torch.Tensor] = None,
) -> torch.Tensor:
    """y = x @ W^T + bias with W dequantised on the fly (reference
    vptq/ops/quant_gemm.py:161-275).  x: [..., in_features] fp16/bf16.

    `prefetch` (extension, optional): a tensor the fused GEMV may read ahead to warm
    L2 / Infinity Cache for the NEXT launch, typically the next layer's `indices`."""
    if vector_quant_dim == "in":
        raise ValueError("Not implemented yet.")
    if not is_indice_packed or residual_indices is not None:
        raise RuntimeError("vptq_amd handles packed int32 indices only (is_indice_packed=True)")
    if x.shape[-1] != in_features:
        raise RuntimeError(f"x has {x.shape[-1]} features, layer expects {in_features}")
    if x.dtype != centroids.dtype:
        raise RuntimeError(f"activation dtype {x.dtype} != weight dtype {centroids.dtype}")
    tokens = x.numel() // x.shape[-1]
    enable_outlier = outlier_centroids is not None and outlier_size > 0
    enable_residual = residual_centroids is not None
    if not x.is_contiguous():
        x = x.contiguous()
    dev = B.require_device(x, bias, indices, centroids, outlier_indices, outlier_centroids,
                           residual_centroids, perm, weight_scale, weight_bias)

    desc = None
    if 1 <= tokens <= B.GEMV_MAX_TOKENS:
        desc, keep = B.make_layer_desc(
            indices=indices, centroids=centroids, res_centroids=residual_centroids,
            outlier_indices=outlier_indices, outlier_centroids=outlier_centroids, perm=perm,
            weight_scale=weight_scale, weight_bias=weight_bias, bias=bias,
            in_features=in_features, out_features=out_features, vector_len=vector_len,
            num_codebooks=num_codebooks, num_centroids=num_centroids,
            num_res_centroids=num_res_centroids if enable_residual else 0,
            group_size=group_size, outlier_size=outlier_size if enable_outlier else 0,
            outlier_vector_len=outlier_vector_len,
            num_outlier_centroids=num_outlier_centroids, prefetch=prefetch)
        if tokens > B.GEMV_ANY_FORMAT_TOKENS and tokens > B.lib().vptq_quant_gemv_max_tokens(desc):
            desc = None  # 9-16 tokens: only the canonical format's GEMV still beats dequant + GEMM
    if desc is not None:
        y = torch.empty(x.shape[:-1] + (out_features,), dtype=x.dtype, device=dev)
        flags = _FLAGS | _safe_flags(indices, centroids, residual_centroids if enable_residual else None, weight_scale, weight_bias,
                                     desc, in_features, out_features)
        with torch.cuda.device(dev):
            sp = B.current_stream_ptr(dev)
            ws, wsb = (None, 0)
            if tokens > 1:   # the one-pass batched-decode kernel of the canonical format wants scratch memory
                ws, wsb = B.gemv_workspace(dev.index if dev.index is not None else torch.cuda.current_device(), sp,
                                           B.lib().vptq_quant_gemv_workspace_bytes(desc, tokens, flags))
            B.check(B.lib().vptq_quant_gemv(desc, x.data_ptr(), y.data_ptr(), tokens, flags,
                                            ws, wsb, sp),
                    "vptq_quant_gemv")
        del keep
        return y

    # many tokens: dequantisation fused into the GEMM where a kernel exists (canonical format)
    if _FUSED_GEMM_MAX_TOKENS >= tokens >= 1 and perm is None:
        fdesc, fkeep = B.make_layer_desc(
            indices=indices, centroids=centroids, res_centroids=residual_centroids,
            outlier_indices=outlier_indices, outlier_centroids=outlier_centroids, perm=None,
            weight_scale=weight_scale, weight_bias=weight_bias, bias=bias,
            in_features=in_features, out_features=out_features, vector_len=vector_len,
            num_codebooks=num_codebooks, num_centroids=num_centroids,
            num_res_centroids=num_res_centroids if enable_residual else 0,
            group_size=group_size, outlier_size=outlier_size if enable_outlier else 0,
            outlier_vector_len=outlier_vector_len, num_outlier_centroids=num_outlier_centroids)
        if B.lib().vptq_quant_gemm_supported(fdesc):
            return quant_gemm_fused(x, fdesc, out_features)

    weight = dequant(
        indices=indices, centroids=centroids, outlier_indices=outlier_indices,
        outlier_centroids=outlier_centroids, res_indices=None,
        res_centroids=residual_centroids, perm=perm, weight_scale=weight_scale,
        weight_bias=weight_bias, is_indice_packed=True, enable_outlier=enable_outlier,
        enable_residual=enable_residual, enable_perm=perm is not None,
        enable_norm=weight_scale is not None and weight_bias is not None,
        num_centroids=num_centroids, num_outlier_centroids=num_outlier_centroids,
        num_res_centroids=num_res_centroids, padding=padding,
        outlier_padding=outlier_padding, num_codebooks=num_codebooks, group_size=group_size,
        outlier_size=outlier_size, vector_len=vector_len,
        outlier_vector_len=outlier_vector_len, vector_quant_dim=vector_quant_dim)
    return F.linear(x, weight, bias)


def quant_gemm_fused(x: torch.Tensor, desc, out_features: int) -> torch.Tensor:
    """y = x @ W^T + bias for any number of tokens with the dequantisation fused into the GEMM
    (`vptq_quant_gemm`: dequantised tile -> LDS -> MFMA; replaces the reference's dequant +
    F.linear, vptq/ops/quant_gemm.py:231-274).  `desc` = a LayerDesc of a supported layer."""
    tokens = x.numel() // x.shape[-1]
    dev = x.device
    y = torch.empty(x.shape[:-1] + (out_features,), dtype=x.dtype, device=dev)
    ws_bytes = B.lib().vptq_quant_gemm_workspace_bytes(desc, tokens)
    ws = torch.empty(max(ws_bytes, 4) // 4, dtype=torch.float32, device=dev) if ws_bytes else None
    with torch.cuda.device(dev):
        B.check(B.lib().vptq_quant_gemm(desc, x.data_ptr(), y.data_ptr(), tokens, _FLAGS,
                                        None if ws is None else ws.data_ptr(), ws_bytes,
                                        B.current_stream_ptr(dev)), "vptq_quant_gemm")
    return y


def quant_gemv_v2(
    x: torch.Tensor,
    bias: Optional[torch.Tensor],
    indices: torch.Tensor,
    centroids: torch.Tensor,
    residual_indices: Optional[torch.Tensor],
    residual_centroids: Optional[torch.Tensor],
    scale_weights: Optional[torch.Tensor],
    scale_bias: Optional[torch.Tensor],
    vector_len: int,
    num_codebooks: int,
    num_centroids: int,
    num_residual_centroids: int,
    out_features: int,
) -> torch.Tensor:
    """Fused dequant + GEMV on the unpacked v2 format (reference
    vptq/ops/quant_gemm.py:278-356; expected result tests/test_quant_gemv.py:49-109).

    x [batch, length, in_features]; indices uint16 [N*I]; residual_indices
    uint8|uint16; centroids [1, k, v]; scale_* [in_features, 1]; bias [1, out]."""
    if num_codebooks != 1:
        raise RuntimeError("quant_gemv_v2 supports one codebook "
                           "(reference csrc/quant_gemv_v2.cu:60)")
    tokens = x.numel() // x.shape[-1]
    if tokens >= 16:
        raise RuntimeError("The input tensor is too large for GEMV to achieve good "
                           "performance. Please use quant_gemm instead.")
    if not x.is_contiguous():
        x = x.contiguous()
    dev = B.require_device(x, bias, indices, centroids, residual_indices, residual_centroids,
                           scale_weights, scale_bias)
    if indices.dtype != torch.uint16 and indices.dtype != torch.int16:
        raise RuntimeError("`indices` must be uint16")
    d = B.V2Desc()
    d.in_features, d.out_features, d.vector_len = x.shape[-1], out_features, vector_len
    d.num_centroids = num_centroids
    has_res = residual_indices is not None and residual_centroids is not None
    d.num_res_centroids = num_residual_centroids if has_res else 0
    d.res_index_bytes = residual_indices.element_size() if has_res else 0
    d.dtype = B.dtype_code(x.dtype)
    d.indices, d.centroids = indices.data_ptr(), centroids.data_ptr()
    d.res_indices = residual_indices.data_ptr() if has_res else None
    d.res_centroids = residual_centroids.data_ptr() if has_res else None
    d.scale_weights = None if scale_weights is None else scale_weights.data_ptr()
    d.scale_bias = None if scale_bias is None else scale_bias.data_ptr()
    d.bias = None if bias is None else bias.data_ptr()
    if indices.numel() != (out_features // vector_len) * x.shape[-1]:
        raise RuntimeError("indices must hold in_features * out_features / vector_len entries")
    y = torch.empty(x.shape[:-1] + (out_features,), dtype=x.dtype, device=dev)
    with torch.cuda.device(dev):
        B.check(B.lib().vptq_quant_gemv_v2(d, x.data_ptr(), y.data_ptr(), tokens, 0,
                                           B.current_stream_ptr(dev)), "vptq_quant_gemv_v2")
    return y
