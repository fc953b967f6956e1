"""Packed-index wire format tools (host side, torch): the format HF VPTQ
checkpoints store in `VQuantLinear.indices`.

Same functions and argument meaning as the reference's
`vptq/utils/pack.py:26-139` (`pack_index`, `unpack_index_tensor`), written
against the format definition rather than its bit-plane expansion:

    indices : int32 [C, N, ceil(G*T/32)],  T = index_bits + res_bits
    each (c, n) row is a little-endian bit stream; element g occupies stream
    bits [g*T, (g+1)*T); value = (res_idx << index_bits) | idx.

These are offline / load-time tools, not part of the per-token hot path (the
kernels read the packed stream directly).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

__all__ = ["pack_index", "unpack_index_tensor", "dtype_convert", "pack_layer_tensors",
           "pack_state_dict", "absorb_perm_layer", "absorb_perm"]


def _as_u16_int64(t: torch.Tensor, index_dtype: torch.dtype) -> torch.Tensor:
    """Read an index tensor stored as uint16 bit patterns (int16 / float16 /
    uint16 storage) as non-negative int64."""
    if t.dtype in (torch.int64, torch.int32) and index_dtype == torch.uint16:
        return t.to(torch.int64) & 0xFFFF
    return t.view(torch.int16).to(torch.int64) & 0xFFFF


def pack_index(
    indice: torch.Tensor,
    index_bits: int,
    res_indice: Optional[torch.Tensor] = None,
    res_bits: int = 0,
    index_dtype: torch.dtype = torch.uint16,
    as_dtype: torch.dtype = torch.int32,
) -> torch.Tensor:
    """(idx [C,N,G], res_idx [C,N,G]) -> int32 [C, N, ceil(G*T/32)]."""
    total_bits = index_bits + res_bits
    assert total_bits <= 32, f"total index bits {total_bits} should be less than 32"
    assert as_dtype in [torch.int32], "as_dtype should be int32"
    merged = _as_u16_int64(indice, index_dtype)
    if res_indice is not None:
        merged = merged | (_as_u16_int64(res_indice, index_dtype) << index_bits)
    *lead, G = merged.shape
    W = (G * total_bits + 31) // 32
    bitpos = torch.arange(G, device=merged.device, dtype=torch.int64) * total_bits
    wi, sh = bitpos >> 5, bitpos & 31
    shifted = merged << sh                       # < 2^63: T <= 32, sh <= 31
    lo = shifted & 0xFFFFFFFF
    hi = shifted >> 32
    flat = torch.zeros((merged.numel() // G, W + 1), dtype=torch.int64, device=merged.device)
    # elements never overlap inside a word, so add == or
    flat.index_add_(1, wi, lo.reshape(-1, G))
    flat.index_add_(1, wi + 1, hi.reshape(-1, G))
    out = flat[:, :W]
    out = torch.where(out >= 2**31, out - 2**32, out).to(torch.int32)
    out = out.reshape(*lead, W)
    # same self-check as the reference (pack.py:69-101): the stream round-trips
    back, rback = unpack_index_tensor(out, index_bits, G, res_bits, G, res_mask_bits=res_bits)
    assert torch.equal(back, merged & ((1 << index_bits) - 1))
    if res_indice is not None:
        assert torch.equal(rback, (merged >> index_bits) & ((1 << res_bits) - 1))
    return out


def unpack_index_tensor(
    packed_tensor: torch.Tensor,
    index_bits: int,
    num_elements: int,
    res_bits: int = 0,
    num_res_elements: int = 0,
    *,
    res_mask_bits: Optional[int] = None,
) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """int32 [C,N,W] -> (idx int64 [C,N,G], res_idx int64 [C,N,G] | None).

    Keeps the reference's behaviour of masking the residual with `index_bits`
    (pack.py:137); identical whenever res_bits <= index_bits.  Tools that REWRITE a
    checkpoint's indices (`absorb_perm_layer`) pass `res_mask_bits=res_bits`, the mask the
    kernels use (reference csrc/kernels/quant_gemv.cuh:120-121), so that a layer with more
    residual than main centroids is not truncated.
    """
    total_bits = index_bits + res_bits
    G = num_elements
    words = packed_tensor.to(torch.int64) & 0xFFFFFFFF
    *lead, W = words.shape
    assert W * 32 >= G * total_bits
    words = torch.cat([words, words.new_zeros(*lead, 1)], dim=-1)
    bitpos = torch.arange(G, device=words.device, dtype=torch.int64) * total_bits
    wi, sh = bitpos >> 5, bitpos & 31
    lo = words[..., wi] >> sh
    hi = (words[..., wi + 1] << (32 - sh)) & 0xFFFFFFFF   # sh == 0 -> shifted out entirely
    hi = torch.where(sh == 0, torch.zeros_like(hi), hi)
    val = (lo | hi) & ((1 << total_bits) - 1)
    indices = val & ((1 << index_bits) - 1)
    res_indices = None
    if res_bits > 0:
        mb = index_bits if res_mask_bits is None else res_mask_bits
        res_indices = (val >> index_bits) & ((1 << mb) - 1)
    return indices, res_indices


def absorb_perm_layer(layer) -> bool:
    """Fold a layer's input permutation into the order of its packed index columns and
    drop `perm` (reference vptq/utils/pack.py:284-394).

    W[o, j] = Wq[o, argsort(perm)[j]] * scale[j] + bias[j]: re-ordering the index columns
    by argsort(perm) gives the same W with the identity permutation; scale / bias are
    indexed by the final column and stay as they are.  Like the reference this only
    applies to single-codebook layers (`group_num == 1`, pack.py:288-293).
    Returns True when the layer was rewritten.
    """
    if not getattr(layer, "enable_perm", False):
        return False
    if layer.group_num > 1 or getattr(layer, "enable_outlier", False):
        return False
    inv = torch.argsort(layer.perm.detach().view(torch.int16).to(torch.int64) & 0xFFFF)
    idx, ridx = unpack_index_tensor(layer.indices.detach(), layer.index_bits, layer.group_size,
                                    layer.res_index_bits, layer.group_size,
                                    res_mask_bits=layer.res_index_bits)
    idx = idx[..., inv]
    if ridx is not None:
        ridx = ridx[..., inv]
    packed = pack_index(idx, layer.index_bits, ridx, layer.res_index_bits, index_dtype=torch.uint16)
    if packed.shape != layer.indices.shape:
        raise ValueError(f"Packed shape {tuple(packed.shape)} doesn't match original shape "
                         f"{tuple(layer.indices.shape)}")
    layer.indices.data = packed.to(layer.indices.device)
    if layer.enable_norm and not hasattr(layer, "norm_dim"):
        layer.norm_dim = 0      # attribute the reference's tools leave behind (pack.py:387-388)
    layer.enable_perm = False
    layer.perm = None
    return True


def absorb_perm(model):
    """absorb_perm_layer over every VQuantLinear of `model`, then clear `enable_perm` in the
    model's `quantization_config` (reference vptq/utils/pack.py:397-433)."""
    from vptq_amd.layers.vqlinear import VQuantLinear
    absorbed = False
    for _, module in model.named_modules():
        if isinstance(module, VQuantLinear):
            absorbed = absorb_perm_layer(module) or absorbed
    qc = getattr(getattr(model, "config", None), "quantization_config", None)
    if absorbed and qc is not None:
        qd = qc if isinstance(qc, dict) else getattr(qc, "__dict__", {})
        for key in ("config_for_layers", "shared_layer_config"):
            for layer_cfg in (qd.get(key) or {}).values():
                if isinstance(layer_cfg, dict) and layer_cfg.get("enable_perm"):
                    layer_cfg["enable_perm"] = False
    return model


# ---------------------------------------------------------------------------------------------
# Offline packing of an UNPACKED checkpoint (what the quantisation algorithm writes) into the
# packed format the kernels read: the reference's `pack_model` / `convert_idx_dtype`
# (vptq/utils/pack.py:142-283).  The reference walks live `VQuantLinear` modules; this build's
# module only exists in packed form (the unpacked mode belongs to the absent quantisation
# algorithm), so the same conversion is offered on the state dict + `quantization_config`.

def dtype_convert(data: torch.Tensor, from_dtype, to_dtype, as_type) -> torch.Tensor:
    """reference pack.py:142-144"""
    return data.view(from_dtype).to(to_dtype).view(as_type)


def _convert(t: torch.Tensor, from_dtype, to_dtype, as_type) -> torch.Tensor:
    # int64 tensors are values, everything else is a bit pattern of `from_dtype`
    # (reference pack.py:158-169)
    return dtype_convert(t, t.dtype if t.dtype == torch.int64 else from_dtype, to_dtype, as_type)


def pack_layer_tensors(tensors: dict, num_centroids: int, num_res_centroids: int,
                       from_dtype=torch.uint16, to_dtype=torch.uint16, as_type=torch.int16) -> dict:
    """One layer: {"indices" [C,N,G], "res_indices" [C,N,G] | None, "outlier_indices" | None,
    "perm" | None, ...} -> the same dict with `indices` packed to int32 [C,N,ceil(G*T/32)],
    `res_indices` removed, `outlier_indices` / `perm` converted to `as_type`.  Other entries
    (codebooks, scale, bias) pass through."""
    out = dict(tensors)
    idx = _convert(tensors["indices"], from_dtype, to_dtype, as_type)
    res = tensors.get("res_indices")
    if res is not None:
        res = _convert(res, from_dtype, to_dtype, as_type)
    for name in ("outlier_indices", "perm"):
        if tensors.get(name) is not None:
            out[name] = _convert(tensors[name], from_dtype, to_dtype, as_type)
    import math
    out["indices"] = pack_index(
        indice=idx, index_bits=int(math.log2(num_centroids)), res_indice=res,
        res_bits=int(math.log2(num_res_centroids)) if res is not None else 0,
        index_dtype=to_dtype)
    out.pop("res_indices", None)
    return out


def pack_state_dict(state: dict, config_for_layers: dict, from_dtype=torch.uint16,
                    to_dtype=torch.uint16, as_type=torch.int16):
    """Whole checkpoint: `state` = flat state dict with unpacked VQuantLinear tensors,
    `config_for_layers` = quantization_config["config_for_layers"] (constructor kwargs per layer
    name).  Returns (new_state, new_config_for_layers) with every listed layer packed and
    `is_indice_packed=True` (reference convert_idx_dtype, pack.py:147-243)."""
    new_state = dict(state)
    new_conf = {}
    for name, conf in config_for_layers.items():
        keys = ("indices", "res_indices", "outlier_indices", "perm")
        layer = {k: state.get(f"{name}.{k}") for k in keys}
        if layer["indices"] is None:
            raise KeyError(f"{name}.indices is not in the state dict")
        k_main = conf["num_centroids"][1]
        k_res = conf["num_res_centroids"][1]
        packed = pack_layer_tensors(layer, k_main, k_res if k_res > 0 else 1, from_dtype,
                                    to_dtype, as_type)
        for k in keys:
            new_state.pop(f"{name}.{k}", None)
            if packed.get(k) is not None:
                new_state[f"{name}.{k}"] = packed[k]
        new_conf[name] = dict(conf, is_indice_packed=True)
    return new_state, new_conf
