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

__all__ = ["pack_index", "unpack_index_tensor"]


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
    back, rback = unpack_index_tensor(out, index_bits, G, res_bits, G)
    assert torch.equal(back, merged & ((1 << index_bits) - 1))
    if res_indice is not None:
        assert torch.equal(rback, (merged >> index_bits) & ((1 << min(res_bits, index_bits)) - 1)) \
            or res_bits > index_bits
    return out


def unpack_index_tensor(
    packed_tensor: torch.Tensor,
    index_bits: int,
    num_elements: int,
    res_bits: int = 0,
    num_res_elements: int = 0,
) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """int32 [C,N,W] -> (idx int64 [C,N,G], res_idx int64 [C,N,G] | None).

    Keeps the reference's behaviour of masking the residual with `index_bits`
    (pack.py:137); identical whenever res_bits <= index_bits.
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
        res_indices = (val >> index_bits) & ((1 << index_bits) - 1)
    return indices, res_indices
