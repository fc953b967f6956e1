#!/usr/bin/env python3
"""Reference goldens at BASELINE sizes (hidden 4096 / 8192), from the REAL reference.

    python tests/golden/gen_golden_big.py        # writes tests/golden/big/*.npz (a few KB each)

A 8192 x 8192 layer holds 16 MiB of packed indices, so nothing but the OUTPUT is stored:
every input tensor (index words, codebooks, scale, bias, x) is procedural - the integer LCG
of _proc.py, rebuilt bit for bit by tests/_cases.py:load_big - and the fixture keeps

* ``y``  = reference ``VQuantLinear.forward(x)`` on its torch CPU path
  (vptq/ops/quant_gemm.py:161-275 -> :43-158 + F.linear),
* the sha256 of ``W`` = reference ``vptq.ops.dequant(...)`` and its first two rows.

Reference run time here: 0.3 s (4096) / 1.3 s (8192) per forward.
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
from _proc import big_tensors  # noqa: E402
import gen_golden as gg  # noqa: E402

CANON = dict(vector_lens=[-1, 8], num_centroids=[-1, 256], num_res_centroids=[-1, 256], group_num=1,
             outlier_size=0, enable_norm=True)
BIG_CASES = [
    # name, I, O, perm, bias, dtype, tokens, dist
    ("h4096_f16_reftest", 4096, 4096, False, False, "f16", 1, "ref-test"),
    ("h4096_f16_llm", 4096, 4096, False, False, "f16", 1, "llm"),
    ("h4096_f16_llm_perm_bias_t2", 4096, 4096, True, True, "f16", 2, "llm"),
    ("h8192_f16_llm", 8192, 8192, False, False, "f16", 1, "llm"),
    ("h8192_bf16_llm", 8192, 8192, False, False, "bf16", 1, "llm"),
    ("h8192_f16_reftest_t4", 8192, 8192, False, False, "f16", 4, "ref-test"),
    # round 3: the many-token routes (the reference's dequant + F.linear branch, vptq/ops/quant_gemm.py:231-274);
    # 256 tokens: every 8th token row of y is stored (+ the last one)
    ("h4096_f16_llm_t64", 4096, 4096, False, False, "f16", 64, "llm"),
    ("h4096_bf16_llm_t64", 4096, 4096, False, False, "bf16", 64, "llm"),
    ("h4096_f16_llm_t256", 4096, 4096, False, False, "f16", 256, "llm"),
    ("h4096_bf16_llm_t256", 4096, 4096, False, False, "bf16", 256, "llm"),
    # round 5: BASELINE configs[3]'s real token count (batch 4 x seq 2048 = 8192 tokens, bf16); every 256th token row of y
    # is stored (+ the last one): 33 rows
    ("h4096_bf16_llm_t8192", 4096, 4096, False, False, "bf16", 8192, "llm"),
]


def stored_rows(tokens):
    """token rows of y a fixture keeps: all of them up to 64 tokens, every 8th + the last up to 1024, every 256th + the last beyond"""
    if tokens <= 64:
        return list(range(tokens))
    return sorted(set(range(0, tokens, 8 if tokens <= 1024 else 256)) | {tokens - 1})


def main():
    vptq = load_reference()
    torch.set_num_threads(1)  # one summation order, whatever the box
    os.makedirs(os.path.join(HERE, "big"), exist_ok=True)
    for ci, (name, I, O, perm, bias, dtype, tokens, dist) in enumerate(BIG_CASES):
        if os.path.exists(os.path.join(HERE, "big", f"{name}.npz")) and "--force" not in sys.argv:
            continue   # (existing fixtures are left alone; --force regenerates them bit for bit)
        seed = 4242 + ci
        dt = gg.TORCH_DT[dtype]
        kw = dict(CANON, enable_perm=perm, bias=bias)
        m = vptq.VQuantLinear(I, O, group_size=I, indices_as_float=False, is_indice_packed=True,
                              dtype=dt, enable_proxy_error=False, **kw)
        t = big_tensors(I, O, 8, 256, 256, perm, bias, tokens, dtype, dist, seed)

        def f(bits_):  # uint16 bit patterns -> tensor of the layer dtype
            return torch.from_numpy(bits_.view(np.int16).copy()).view(dt)

        m.indices.data = torch.from_numpy(t["indices"].copy()).reshape(m.indices.shape)
        m.centroids.weight.data = f(t["centroids"]).reshape(m.centroids.weight.shape)
        m.res_centroids.weight.data = f(t["res_centroids"]).reshape(m.res_centroids.weight.shape)
        m.weight_scale.data = f(t["weight_scale"])
        m.weight_bias.data = f(t["weight_bias"])
        if perm:
            m.perm.data = torch.from_numpy(t["perm"].view(np.int16).copy())
        if bias:
            m.bias.data = f(t["bias"])
        x = f(t["x"]).reshape(1, tokens, I)
        m.eval()
        with torch.no_grad():
            W = gg.ref_dequant(vptq, m)
            y = m(x)
        Wb = gg.bits(W)
        cfg = dict(name=name, in_features=I, out_features=O, dtype=dtype, tokens=tokens, dist=dist,
                   seed=seed, perm=perm, bias=bias, vector_len=8, num_centroids=256,
                   num_res_centroids=256, W_sha256=hashlib.sha256(Wb.tobytes()).hexdigest(),
                   torch=torch.__version__)
        out = os.path.join(HERE, "big", f"{name}.npz")
        np.savez_compressed(out, config=np.frombuffer(json.dumps(cfg).encode(), dtype=np.uint8),
                            y=gg.bits(y)[:, stored_rows(tokens), :], W_head=Wb[:2].copy())
        print(f"{name}: y{tuple(y.shape)} -> {os.path.getsize(out) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
