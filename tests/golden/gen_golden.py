#!/usr/bin/env python3
"""Generate golden vectors from the REAL reference (run in the build container).

    python tests/golden/gen_golden.py            # writes tests/golden/*.npz

Imports microsoft/VPTQ read-only from /root/reference (two shims, see
_refshim.py), builds reference ``VQuantLinear`` modules on CPU with seeded
random parameters, and records

* every parameter tensor (raw bit patterns),
* ``W``  = reference ``vptq.ops.dequant(...)``  (sha256 of the bit pattern, the
  first 16 rows verbatim),
* ``y``  = reference ``VQuantLinear.forward(x)`` on the torch fallback
  (vptq/ops/quant_gemm.py:247-274).

``/root/reference`` does not exist on the GPU box, so these fixtures are what
pins the oracle (tests/test_oracle_golden.py) and, through it, the HIP path.
"""
import hashlib
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from _refshim import load_reference  # noqa: E402
from _proc import PROC_MIN_ELEMS, proc_values  # noqa: E402

CASES = [
    # name, I, O, ctor kwargs, dtype, tokens, dist
    ("canon_k256x2", 512, 256, dict(vector_lens=[-1, 8], num_centroids=[-1, 256], num_res_centroids=[-1, 256], group_num=1, outlier_size=0, enable_norm=True, enable_perm=False, bias=False), "f16", 1, "ref-test"),
    ("canon_llm", 512, 256, dict(vector_lens=[-1, 8], num_centroids=[-1, 256], num_res_centroids=[-1, 256], group_num=1, outlier_size=0, enable_norm=True, enable_perm=False, bias=False), "f16", 1, "llm"),
    ("canon_perm_bias_t2", 512, 256, dict(vector_lens=[-1, 8], num_centroids=[-1, 256], num_res_centroids=[-1, 256], group_num=1, outlier_size=0, enable_norm=True, enable_perm=True, bias=True), "f16", 2, "ref-test"),
    ("canon_bf16", 512, 256, dict(vector_lens=[-1, 8], num_centroids=[-1, 256], num_res_centroids=[-1, 256], group_num=1, outlier_size=0, enable_norm=True, enable_perm=False, bias=False), "bf16", 1, "llm"),
    ("k4096_v6_pad", 384, 250, dict(vector_lens=[-1, 6], num_centroids=[-1, 4096], num_res_centroids=[-1, -1], group_num=1, outlier_size=0, enable_norm=True, enable_perm=False, bias=False), "f16", 1, "llm"),
    ("k65536_r256_t24", 320, 128, dict(vector_lens=[-1, 8], num_centroids=[-1, 65536], num_res_centroids=[-1, 256], group_num=1, outlier_size=0, enable_norm=True, enable_perm=True, bias=False), "f16", 1, "llm"),
    ("k8192_r256_t21", 352, 128, dict(vector_lens=[-1, 8], num_centroids=[-1, 8192], num_res_centroids=[-1, 256], group_num=1, outlier_size=0, enable_norm=True, enable_perm=False, bias=True), "f16", 1, "ref-test"),
    ("k65536_nores_nonorm", 256, 128, dict(vector_lens=[-1, 8], num_centroids=[-1, 65536], num_res_centroids=[-1, -1], group_num=1, outlier_size=0, enable_norm=False, enable_perm=False, bias=False), "f16", 1, "llm"),
    ("v12_k4096x2", 288, 132, dict(vector_lens=[-1, 12], num_centroids=[-1, 4096], num_res_centroids=[-1, 4096], group_num=1, outlier_size=0, enable_norm=True, enable_perm=False, bias=False), "f16", 1, "llm"),
    ("v16_k65536x2_t32", 288, 128, dict(vector_lens=[-1, 16], num_centroids=[-1, 65536], num_res_centroids=[-1, 65536], group_num=1, outlier_size=0, enable_norm=True, enable_perm=False, bias=False), "bf16", 1, "llm"),
    ("v4_k256_nores", 264, 64, dict(vector_lens=[-1, 4], num_centroids=[-1, 256], num_res_centroids=[-1, -1], group_num=1, outlier_size=0, enable_norm=True, enable_perm=False, bias=False), "f16", 3, "ref-test"),
    ("c2_outlier_perm_bias_pad", 16 + 2 * 160, 44, dict(vector_lens=[4, 8], num_centroids=[64, 256], num_res_centroids=[-1, 256], group_num=2, outlier_size=16, enable_norm=True, enable_perm=True, bias=True), "f16", 1, "ref-test"),
    ("c4_k1024_r16", 4 * 96, 96, dict(vector_lens=[-1, 8], num_centroids=[-1, 1024], num_res_centroids=[-1, 16], group_num=4, outlier_size=0, enable_norm=True, enable_perm=False, bias=False), "f16", 5, "llm"),
    # several tokens through the canonical format (the 2-4-token kernels; 16 tokens = launches of 4)
    ("canon_t4_bias", 2048, 96, dict(vector_lens=[-1, 8], num_centroids=[-1, 256], num_res_centroids=[-1, 256], group_num=1, outlier_size=0, enable_norm=True, enable_perm=False, bias=True), "f16", 4, "llm"),
    ("canon_t16_perm", 1024, 136, dict(vector_lens=[-1, 8], num_centroids=[-1, 256], num_res_centroids=[-1, 256], group_num=1, outlier_size=0, enable_norm=True, enable_perm=True, bias=False), "f16", 16, "ref-test"),
    ("canon_bf16_t2", 4104, 64, dict(vector_lens=[-1, 8], num_centroids=[-1, 256], num_res_centroids=[-1, 256], group_num=1, outlier_size=0, enable_norm=True, enable_perm=False, bias=False), "bf16", 2, "llm"),
]

TORCH_DT = {"f16": torch.float16, "bf16": torch.bfloat16}


def _rand(gen, shape, mean, std, dt):
    return (torch.randn(shape, generator=gen) * std + mean).to(dt)


def _codebook(gen, shape, mean, std, dt, proc, key, seed):
    """Small codebooks: torch.randn (stored).  Large: procedural, not stored."""
    n = int(np.prod(shape))
    if n < PROC_MIN_ELEMS:
        return _rand(gen, shape, mean, std, dt)
    proc[key] = dict(seed=seed, scale=float(3 * std), mean=float(mean))
    v = proc_values(n, seed, 3 * std) + np.float32(mean)
    return torch.from_numpy(v).reshape(shape).to(dt)


def bits(t: torch.Tensor) -> np.ndarray:
    if t.dtype in (torch.float16, torch.bfloat16, torch.int16):
        return t.detach().contiguous().view(torch.int16).numpy().view(np.uint16).copy()
    return t.detach().contiguous().numpy().copy()


def build(vptq, name, I, O, kw, dtype, tokens, dist, seed):
    dt = TORCH_DT[dtype]
    gen = torch.Generator().manual_seed(seed)
    C = kw["group_num"]
    S = kw["outlier_size"]
    G = (I - S) // C
    m = vptq.VQuantLinear(I, O, group_size=G, indices_as_float=False, is_indice_packed=True,
                          dtype=dt, enable_proxy_error=False, **kw)
    p = (dict(c=(0.02, 0.5), r=(0.02, 0.5), s=(0.02, 0.5), b=(0.02, 0.5), x=(0.02, 0.5), o=(0.02, 0.5))
         if dist == "ref-test" else
         dict(c=(0.0, 0.02), r=(0.0, 0.005), s=(1.0, 0.1), b=(0.0, 0.01), x=(0.0, 1.0), o=(0.0, 0.02)))
    m.indices.data = torch.randint(-2**31, 2**31 - 1, m.indices.shape, generator=gen, dtype=torch.int64).to(torch.int32)
    proc = {}
    m.centroids.weight.data = _codebook(gen, m.centroids.weight.shape, *p["c"], dt, proc, "centroids", seed * 10 + 1)
    if m.enable_residual:
        m.res_centroids.weight.data = _codebook(gen, m.res_centroids.weight.shape, *p["r"], dt, proc, "res_centroids", seed * 10 + 2)
    if m.enable_outlier:
        m.outlier_centroids.weight.data = _rand(gen, m.outlier_centroids.weight.shape, *p["c"], dt)
        oi = torch.randint(0, m.num_outlier_centroids, m.outlier_indices.shape, generator=gen)
        m.outlier_indices.data = oi.to(torch.uint16).view(torch.int16)
    if m.enable_perm:
        m.perm.data = torch.randperm(I, generator=gen).to(torch.uint16).view(torch.int16)
    if m.enable_norm:
        m.weight_scale.data = _rand(gen, (I,), *p["s"], dt)
        m.weight_bias.data = _rand(gen, (I,), *p["b"], dt)
    if m.bias is not None:
        m.bias.data = _rand(gen, (O,), *p["o"], dt)
    x = _rand(gen, (1, tokens, I), *p["x"], dt)
    return m.eval(), x, proc


def ref_dequant(vptq, m):
    return vptq.ops.dequant(
        indices=m.indices, centroids=m.centroids.weight.view(m.num_codebooks, m.num_centroids, m.vector_len),
        outlier_indices=m.outlier_indices,
        outlier_centroids=m.outlier_centroids.weight if m.enable_outlier else None,
        res_indices=None,
        res_centroids=m.res_centroids.weight if m.enable_residual else None,
        perm=getattr(m, "perm", None), weight_scale=m.weight_scale, weight_bias=m.weight_bias,
        is_indice_packed=True, enable_outlier=m.enable_outlier, enable_residual=m.enable_residual,
        enable_perm=m.enable_perm, enable_norm=m.enable_norm, num_centroids=m.num_centroids,
        num_outlier_centroids=m.num_outlier_centroids, num_res_centroids=m.num_res_centroids,
        padding=m.padding, outlier_padding=m.outlier_padding, num_codebooks=m.num_codebooks,
        group_size=m.group_size, outlier_size=m.outlier_size, vector_len=m.vector_len,
        outlier_vector_len=m.outlier_vector_len)


def main():
    vptq = load_reference()
    torch.set_num_threads(1)
    for ci, (name, I, O, kw, dtype, tokens, dist) in enumerate(CASES):
        m, x, proc = build(vptq, name, I, O, kw, dtype, tokens, dist, seed=1234 + ci)
        with torch.no_grad():
            W = ref_dequant(vptq, m)
            y = m(x)
        assert W.shape == (O, I) and y.shape == (1, tokens, O)
        Wb = bits(W)
        cfg = dict(name=name, in_features=I, out_features=O, dtype=dtype, tokens=tokens, dist=dist,
                   vector_len=m.vector_len, num_centroids=m.num_centroids,
                   num_res_centroids=m.num_res_centroids, num_codebooks=m.num_codebooks,
                   group_size=m.group_size, outlier_size=m.outlier_size,
                   outlier_vector_len=m.outlier_vector_len,
                   num_outlier_centroids=m.num_outlier_centroids,
                   enable_norm=m.enable_norm, enable_perm=m.enable_perm, has_bias=m.bias is not None,
                   ctor=kw, proc=proc, W_sha256=hashlib.sha256(Wb.tobytes()).hexdigest(),
                   torch=torch.__version__)
        arrs = dict(config=np.frombuffer(json.dumps(cfg).encode(), dtype=np.uint8),
                    indices=bits(m.indices), x=bits(x), y=bits(y), W_head=Wb[:16].copy())
        if "centroids" not in proc:
            arrs["centroids"] = bits(m.centroids.weight)
        if m.enable_residual and "res_centroids" not in proc:
            arrs["res_centroids"] = bits(m.res_centroids.weight)
        if m.enable_outlier:
            arrs["outlier_indices"] = bits(m.outlier_indices)
            arrs["outlier_centroids"] = bits(m.outlier_centroids.weight)
        if m.enable_perm:
            arrs["perm"] = bits(m.perm)
        if m.enable_norm:
            arrs["weight_scale"] = bits(m.weight_scale)
            arrs["weight_bias"] = bits(m.weight_bias)
        if m.bias is not None:
            arrs["bias"] = bits(m.bias)
        out = os.path.join(HERE, f"{name}.npz")
        np.savez_compressed(out, **arrs)
        print(f"{name}: W{tuple(W.shape)} y{tuple(y.shape)} -> {os.path.getsize(out)/1024:.0f} KiB")


if __name__ == "__main__":
    main()
