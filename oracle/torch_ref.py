"""Torch restatement of the reference's CPU forward - TEST INFRASTRUCTURE, and the
`cpu_baseline` leg of bench.py (never imported by the product).

What the reference runs when its native extension is absent (vptq/ops/quant_gemm.py:28-40 ->
:231-274): `dequant` (:43-158) builds the dense W with torch tensor ops - the packed words are
blown up to one element per BIT (`unpack_index_tensor`, vptq/utils/pack.py:105-139), re-summed
to indices, both codebooks are gathered, added, transposed, scaled and biased, each op rounding
to the 16-bit dtype - and `F.linear` contracts.  This file performs the same sequence of
tensor ops (same intermediate shapes, hence the same memory traffic and the same torch CPU
kernels) for the single-codebook, no-outlier configurations bench.py times; it is pinned
against the real reference's goldens in tests/test_oracle_golden.py.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def unpack_bits(packed: torch.Tensor, index_bits: int, group_size: int, res_bits: int):
    """int32 [C, N, W] -> (idx, ridx) int64 [C, N, G] through a one-element-per-bit tensor
    (pack.py:113-139).  The residual is masked with `index_bits`, like the reference (:137)."""
    T = index_bits + res_bits
    shifts = torch.arange(32, device=packed.device).view(1, 1, 1, 32)
    bits = (packed.unsqueeze(-1) >> shifts) & 1                      # [C, N, W, 32]
    bits = bits.reshape(*packed.shape[:-1], -1)[..., :group_size * T]
    bits = bits.reshape(*packed.shape[:-1], group_size, T)
    weights = torch.arange(T, device=packed.device).view(1, 1, 1, T)
    val = (bits << weights).sum(dim=-1).to(torch.int64)              # [C, N, G]
    idx = val & ((1 << index_bits) - 1)
    ridx = ((val >> index_bits) & ((1 << index_bits) - 1)) if res_bits > 0 else None
    return idx, ridx


def _lookup(codebook: torch.Tensor, idx: torch.Tensor, C: int, G: int, v: int) -> torch.Tensor:
    """codebook [C, k, v], idx [C, N, G] -> [N * v, C * G] (quant_gemm.py:92-104)."""
    flat = idx.unsqueeze(-1).expand(-1, -1, -1, v).reshape(C, -1, v)
    sel = torch.gather(codebook, 1, flat).view(C, -1, G, v)          # [C, N, G, v]
    return sel.permute(0, 1, 3, 2).reshape(C, -1, G).permute(1, 0, 2).reshape(-1, C * G)


def dequant(indices, centroids, res_centroids, weight_scale, weight_bias, *, num_centroids,
            num_res_centroids, vector_len, group_size, out_features, num_codebooks=1):
    C, G, v = num_codebooks, group_size, vector_len
    ib = math.ceil(math.log2(num_centroids))
    rb = math.ceil(math.log2(num_res_centroids)) if num_res_centroids > 0 else 0
    idx, ridx = unpack_bits(indices, ib, G, rb)
    W = _lookup(centroids.view(C, num_centroids, v), idx, C, G, v)
    if rb:
        W = W + _lookup(res_centroids.view(C, num_res_centroids, v), ridx, C, G, v)
    W = W[:out_features]                                             # strip the row padding
    if weight_scale is not None:
        W = W * weight_scale + weight_bias
    return W


def forward(x, indices, centroids, res_centroids, weight_scale, weight_bias, bias=None, **kw):
    return F.linear(x, dequant(indices, centroids, res_centroids, weight_scale, weight_bias, **kw), bias)
