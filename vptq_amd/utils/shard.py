"""Tensor-parallel shards of a VQuantLinear (BASELINE config #5; nothing comparable exists in
the reference, whose only multi-GPU mode is accelerate's sequential layer placement,
vptq/layers/model_base.py:186-194).

Two ways to cut ONE layer, both pure slicing of the reference-format tensors:

* `shard_out_features` — split the vector-rows `n` (Megatron "column parallel"): every rank
  produces a disjoint slice of y; combine with an all-gather.  Exact.
* `shard_in_features` — split the input columns `g` (Megatron "row parallel"): each row's bit
  stream is cut on a 32-bit word boundary, scale / bias follow the columns, every rank produces
  a full-length PARTIAL y; combine with an all-reduce (sum) and add `bias` once.  Needs the
  permutation absorbed first (`absorb_perm`), one codebook, no outlier columns.

At batch 1 the exchange is an 8-16 KB message, i.e. latency-bound (~10-20 us on xGMI) while
the sharded kernel shrinks to ~1-3 us: this cannot scale and is provided for completeness and
for larger batches; independent layers per GPU (bench.py's default) is the mode that scales.
"""
from __future__ import annotations

import torch

from vptq_amd.layers.vqlinear import VQuantLinear


def _split(n: int, rank: int, world: int):
    """contiguous, near-even split of range(n)"""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _clone_like(layer: VQuantLinear, in_features: int, out_features: int, group_size: int,
                enable_perm: bool, bias: bool) -> VQuantLinear:
    dev, dt = layer.indices.device, layer.centroids.weight.dtype
    with torch.device("meta"):
        new = VQuantLinear(
            in_features, out_features,
            vector_lens=[layer.outlier_vector_len, layer.vector_len],
            num_centroids=[layer.num_outlier_centroids, layer.num_centroids],
            num_res_centroids=[-1, layer.num_res_centroids if layer.enable_residual else -1],
            group_num=layer.group_num, group_size=group_size, outlier_size=layer.outlier_size,
            indices_as_float=layer.indices_as_float, enable_norm=layer.enable_norm,
            enable_perm=enable_perm, is_indice_packed=True, bias=bias, dtype=dt,
            enable_proxy_error=False)
    new = new.to_empty(device=dev)
    # codebooks are replicated (shared storage, no copy)
    new.centroids.weight = layer.centroids.weight
    if layer.enable_residual:
        new.res_centroids.weight = layer.res_centroids.weight
    if layer.enable_outlier:
        new.outlier_centroids.weight = layer.outlier_centroids.weight
    return new.eval()


def shard_out_features(layer: VQuantLinear, rank: int, world: int) -> VQuantLinear:
    """Rank `rank`'s slice of the vector-rows: y_rank = y[..., n0*v : n1*v]."""
    v, N = layer.vector_len, layer.num_indices
    n0, n1 = _split(N, rank, world)
    if n1 == n0:
        raise ValueError(f"layer has {N} vector-rows, cannot give rank {rank} of {world} any")
    o0, o1 = n0 * v, min(n1 * v, layer.out_features)
    if layer.enable_outlier and (o0 % layer.outlier_vector_len or
                                 (o1 % layer.outlier_vector_len and o1 != layer.out_features)):
        raise ValueError("outlier vectors straddle the shard boundary")
    new = _clone_like(layer, layer.in_features, o1 - o0, layer.group_size, layer.enable_perm,
                      layer.bias is not None)
    new.indices.data = layer.indices.data[:, n0:n1, :].contiguous()
    if layer.enable_outlier:
        ov = layer.outlier_vector_len
        new.outlier_indices.data = layer.outlier_indices.data[:, o0 // ov:(o1 + ov - 1) // ov, :].contiguous()
    if layer.enable_perm:
        new.perm = layer.perm
    if layer.enable_norm:
        new.weight_scale, new.weight_bias = layer.weight_scale, layer.weight_bias
    if layer.bias is not None:
        new.bias.data = layer.bias.data[o0:o1].contiguous()
    new.shard = ("out", o0, o1)
    return new


def shard_in_features(layer: VQuantLinear, rank: int, world: int) -> VQuantLinear:
    """Rank `rank`'s slice of the input columns: y = sum_rank forward_rank(x[..., g0:g1])
    (+ bias, carried by rank 0 only)."""
    if layer.enable_perm:
        raise ValueError("absorb the permutation first (vptq_amd.utils.pack.absorb_perm_layer)")
    if layer.group_num != 1 or layer.enable_outlier:
        raise ValueError("row-parallel shards need one codebook and no outlier columns")
    G, T = layer.group_size, layer.total_index_bits
    if G % world:
        raise ValueError(f"group_size {G} not divisible by world {world}")
    Gs = G // world
    if (Gs * T) % 32:
        raise ValueError(f"a shard of {Gs} columns x {T} bits does not end on a 32-bit word")
    g0 = rank * Gs
    w0, w1 = g0 * T // 32, (g0 + Gs) * T // 32
    new = _clone_like(layer, Gs, layer.out_features, Gs, False,
                      layer.bias is not None and rank == 0)
    new.indices.data = layer.indices.data[:, :, w0:w1].contiguous()
    if layer.enable_norm:
        new.weight_scale.data = layer.weight_scale.data[g0:g0 + Gs].contiguous()
        new.weight_bias.data = layer.weight_bias.data[g0:g0 + Gs].contiguous()
    if new.bias is not None:
        new.bias = layer.bias
    new.shard = ("in", g0, g0 + Gs)
    return new


def forward_partial_f32(layer: VQuantLinear, x: torch.Tensor) -> torch.Tensor:
    """`layer(x)` with the output left in float32, i.e. BEFORE the one rounding of the
    reference's `F.linear` (vptq/ops/quant_gemm.py:274): `VPTQ_GEMV_OUT_F32` of the fused
    GEMV.  What a row-parallel rank contributes to the all-reduce.  Any token count."""
    from vptq_amd import _backend as B
    from vptq_amd import ops
    xc = layer._check_activation(x)
    tokens = xc.numel() // xc.shape[-1]
    if tokens < 1:
        raise RuntimeError("forward_partial_f32 needs at least one token")
    cache = layer._descriptor()
    _, desc, _, dev, fn = cache[:5]
    # what vptq_quant_gemv takes for THIS layer: 64 tokens for fp16 layers of the canonical format
    # (vptq_quant_gemv_max_tokens answers 48 for them), 16 for every other layer; beyond that the partial
    # sum comes from the dense route: dequant + a matmul that accumulates and stays in fp32
    limit = B.GEMV_MAX_TOKENS if cache[5] >= 48 else 16

    def dense():
        W = layer.dequant().float()
        y = torch.matmul(xc.float(), W.t())
        if layer.bias is not None:
            y = y + layer.bias.float()
        return y

    if tokens > limit:
        return dense()
    y = torch.empty(xc.shape[:-1] + (layer.out_features,), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        sp = torch.cuda.current_stream(dev).cuda_stream
        ws, wsb = B.gemv_workspace(cache[8], sp, cache[10]) if tokens > 1 else (None, 0)
        rc = fn(desc, xc.data_ptr(), y.data_ptr(), tokens,
                ops.quant_gemm_flags() | B.GEMV_OUT_F32 | cache[9], ws, wsb, sp)
    if rc == B.E_TOKENS and tokens > B.GEMV_ANY_FORMAT_TOKENS:
        # the fused path takes 17+ tokens of this layer only under run-time conditions the descriptor cannot
        # promise (a workspace - none is handed out inside a stream capture before one exists -, no exact-arithmetic
        # flag, 16-byte aligned activations): the dense route, as VQuantLinear._gemv_cached does
        return dense()
    B.check(rc, "vptq_quant_gemv")
    return y


def row_parallel_forward(shard: VQuantLinear, x_full: torch.Tensor, group=None) -> torch.Tensor:
    """One row-parallel (input-column sharded) layer step on this rank: slice x, fused GEMV
    with fp32 partial output, all-reduce (RCCL over xGMI when the process group is "nccl"),
    ONE rounding to the activation dtype - the single exchange step of BASELINE config #5
    (SURVEY.md 8e).  `shard` = shard_in_features(layer, rank, world); its output bias lives on
    rank 0 and rides inside rank 0's partial sum."""
    import torch.distributed as dist
    g0, g1 = shard.shard[1], shard.shard[2]
    part = forward_partial_f32(shard, x_full[..., g0:g1])
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(part, op=dist.ReduceOp.SUM, group=group)
    return part.to(x_full.dtype)
