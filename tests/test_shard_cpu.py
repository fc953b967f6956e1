"""Tensor-parallel shards of one layer: slicing checked with the oracle as the compute
stand-in (CPU), and the exchange step exercised with world_size 2 on gloo."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

from oracle import vptq_oracle as vo  # noqa: E402
from _gpu_util import spec_to_module, module_to_spec  # noqa: E402
from vptq_amd.utils.shard import shard_in_features, shard_out_features  # noqa: E402


def _layer(**kw):
    return vo.make_layer(512, 136, dist="llm", seed=21, bias=True, **kw)


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_out_feature_shards_concatenate_exactly(world):
    L = _layer(enable_perm=True)
    m = spec_to_module(L, "cpu")
    x = vo.from_f32(np.random.default_rng(0).standard_normal((1, 2, 512)).astype(np.float32), "f16")
    full = vo.forward(L, x)
    parts = [vo.forward(module_to_spec(shard_out_features(m, r, world)), x) for r in range(world)]
    assert (np.concatenate(parts, axis=-1) == full).all()
    assert sum(p.shape[-1] for p in parts) == 136


@pytest.mark.parametrize("world", [1, 2, 4])
def test_in_feature_shards_sum_to_full(world):
    L = _layer()
    m = spec_to_module(L, "cpu")
    x = vo.from_f32(np.random.default_rng(1).standard_normal((1, 1, 512)).astype(np.float32), "f16")
    full = vo.to_f32(vo.forward(L, x), "f16")
    acc = np.zeros_like(full, dtype=np.float64)
    for r in range(world):
        s = shard_in_features(m, r, world)
        g0, g1 = s.shard[1], s.shard[2]
        Ls = module_to_spec(s)
        assert (Ls.bias is not None) == (r == 0)
        W = vo.to_f32(vo.dequant(Ls), "f16").astype(np.float64)
        y = vo.to_f32(x[..., g0:g1], "f16").astype(np.float64) @ W.T          # fp32-class partial
        if Ls.bias is not None:
            y = y + vo.to_f32(Ls.bias, "f16")
        acc += y
    assert np.abs(acc - full).max() / np.abs(full).max() <= 1e-3
    # the shards' dense weights tile the full weight exactly
    Wfull = vo.dequant(L)
    Wcat = np.concatenate([vo.dequant(module_to_spec(shard_in_features(m, r, world)))
                           for r in range(world)], axis=1)
    assert (Wcat == Wfull).all()


def test_in_feature_shard_preconditions():
    with pytest.raises(ValueError, match="absorb"):
        shard_in_features(spec_to_module(_layer(enable_perm=True), "cpu"), 0, 2)
    with pytest.raises(ValueError, match="32-bit word"):
        shard_in_features(spec_to_module(vo.make_layer(24, 16, num_centroids=4096, num_res_centroids=0, seed=1), "cpu"), 0, 8)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    L = _layer()
    m = spec_to_module(L, "cpu")
    x = vo.from_f32(np.random.default_rng(2).standard_normal((1, 1, 512)).astype(np.float32), "f16")
    # column parallel: all-gather of disjoint output slices (uneven split: 136/8 = 17 rows)
    so = shard_out_features(m, rank, world)
    y_part = torch.from_numpy(vo.to_f32(vo.forward(module_to_spec(so), x), "f16")).reshape(-1)
    sizes = [shard_out_features(m, r, world).out_features for r in range(world)]
    # uneven shards: pad every slice to the largest one for the collective, then trim
    mx = max(sizes)
    padded = torch.zeros(mx); padded[:y_part.numel()] = y_part
    bufs = [torch.empty(mx) for _ in sizes]
    dist.all_gather(bufs, padded)
    y_col = torch.cat([b[:n] for b, n in zip(bufs, sizes)])
    # row parallel: all-reduce (sum) of full-length partials
    si = shard_in_features(m, rank, world)
    Ls = module_to_spec(si)
    W = vo.to_f32(vo.dequant(Ls), "f16").astype(np.float64)
    part = vo.to_f32(x[..., si.shard[1]:si.shard[2]], "f16").astype(np.float64) @ W.T
    if Ls.bias is not None:
        part = part + vo.to_f32(Ls.bias, "f16")
    t = torch.from_numpy(part.reshape(-1))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    full = torch.from_numpy(vo.to_f32(vo.forward(L, x), "f16").reshape(-1))
    q.put((rank, bool(torch.equal(y_col, full)),
           float((t.float() - full).abs().max() / full.abs().max())))
    dist.destroy_process_group()


def test_tp_exchange_world2_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=90) for _ in range(world)]
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, col_exact, row_err in res:
        assert col_exact and row_err <= 1e-3


def _worker_rpf(rank, world, port, q):
    """`row_parallel_forward` ITSELF on this rank's shard (the function bench.py's multi-GPU step is built from), with a
    CPU stand-in for the one GPU call inside it (`forward_partial_f32`: the fused GEMV with float32 output)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import vptq_amd.utils.shard as shard_mod

    def partial_f32_on_cpu(layer, x):     # what the kernel computes: fp32 sums of this shard's columns (+ bias on rank 0)
        Ls = module_to_spec(layer)
        W = vo.to_f32(vo.dequant(Ls), "f16").astype(np.float64)
        xb = x.contiguous().view(torch.int16).numpy().view(np.uint16)
        part = vo.to_f32(xb, "f16").astype(np.float64) @ W.T
        if Ls.bias is not None:
            part = part + vo.to_f32(Ls.bias, "f16")
        return torch.from_numpy(part.astype(np.float32))

    shard_mod.forward_partial_f32 = partial_f32_on_cpu
    L = _layer()
    m = spec_to_module(L, "cpu")
    xb = vo.from_f32(np.random.default_rng(3).standard_normal((1, 2, 512)).astype(np.float32), "f16")
    x = torch.from_numpy(xb.view(np.int16).copy()).view(torch.float16)
    si = shard_in_features(m, rank, world)
    assert (si.bias is not None) == (rank == 0 and m.bias is not None)     # the output bias rides on rank 0 only
    y = shard_mod.row_parallel_forward(si, x)                               # slice x, partial sums, all-reduce, ONE rounding
    want = vo.forward(L, xb)
    got = y.contiguous().view(torch.int16).numpy().view(np.uint16)
    ident = float((got.reshape(-1) == np.asarray(want).reshape(-1)).mean())
    err = float(np.abs(vo.to_f32(got, "f16") - vo.to_f32(want, "f16")).max() / np.abs(vo.to_f32(want, "f16")).max())
    q.put((rank, y.dtype == torch.float16 and tuple(y.shape) == (1, 2, m.out_features), ident, err))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_row_parallel_forward_world_gloo(world):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker_rpf, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, shape_ok, ident, err in res:
        # every rank holds the SAME reduced result; fp32 partial sums + one rounding = the reference's F.linear model
        assert shape_ok and ident >= 0.98 and err <= 1e-3, (rank, ident, err)
