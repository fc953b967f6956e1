#!/usr/bin/env python3
"""Reference goldens for the NON-canonical formats at the sizes and token counts where their own kernels run
(tall layers of the LDS-resident formats, 5-8 tokens of the k = 65536 formats, vector lengths 2 / 4 / 6 / 10),
from the REAL reference.

    python tests/golden/gen_golden_fmt.py        # writes tests/golden/fmt/*.npz (a few KB each)

Same scheme as gen_golden_big.py: every input tensor is procedural (tests/golden/_proc.py:big_tensors, rebuilt
bit for bit by tests/_cases.py:load_fmt); the fixture keeps ``y`` = reference ``VQuantLinear.forward(x)`` on its
torch CPU path (vptq/ops/quant_gemm.py:161-275), the sha256 of ``W`` = reference ``vptq.ops.dequant`` and its
first two rows.
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

FMT_CASES = [
    # name, I, O, v, k, kr, perm, bias, dtype, tokens, dist
    ("t1_k8192_r256_8192x1024", 1024, 8192, 8, 8192, 256, False, False, "f16", 1, "llm"),
    ("t1_k4096_r512_perm_bias", 2048, 8192 + 8, 8, 4096, 512, True, True, "f16", 1, "ref-test"),
    ("t1_k8192_r256_bf16", 1024, 8192, 8, 8192, 256, False, False, "bf16", 1, "llm"),
    ("t8_k65536_r256", 2048, 512, 8, 65536, 256, False, True, "f16", 8, "llm"),
    ("t5_k65536_r65536_perm", 1024, 256, 8, 65536, 65536, True, False, "f16", 5, "llm"),
    ("t7_v6_k4096_r16", 1032, 300, 6, 4096, 16, False, False, "f16", 7, "llm"),
    ("t1_v10_k4096_r256", 1024, 1000, 10, 4096, 256, False, True, "f16", 1, "ref-test"),
    ("t2_v2_k256_r16_perm", 512, 258, 2, 256, 16, True, False, "f16", 2, "llm"),
    ("t8_v4_k4096_r256_bf16", 1024, 260, 4, 4096, 256, False, False, "bf16", 8, "llm"),
    # round 3: v8-k65536-0 (no residual codebook), one token: the layers of the sliced layout (gemv_sliced.hip)
    ("t1_k65536_r0_4096x4096", 4096, 4096, 8, 65536, 0, False, True, "f16", 1, "llm"),
    ("t1_k65536_r0_bf16", 2048, 8192 + 8, 8, 65536, 0, False, False, "bf16", 1, "llm"),
    # v8-k65536-256 (the format of most published checkpoints), one token
    ("t1_k65536_r256_4096x4096", 4096, 4096, 8, 65536, 256, False, True, "f16", 1, "llm"),
    ("t1_k65536_r256_bf16", 2048, 4096 + 8, 8, 65536, 256, False, False, "bf16", 1, "llm"),
    # round 4: v8-k65536-65536 (T = 32, the "4 bit" format of every published model family), one token: two passes of the
    # sliced kernel, one per table
    ("t1_k65536_r65536_4096x4096", 4096, 4096, 8, 65536, 65536, False, True, "f16", 1, "llm"),
    ("t1_k65536_r65536_bf16_perm", 2048, 2048 + 8, 8, 65536, 65536, True, False, "bf16", 1, "llm"),
    # vector length 16: v16-k65536-65536 ("2 bits" of most model families) and v16-k65536-0
    ("t1_v16_k65536_r65536_4096x4096", 4096, 4096, 16, 65536, 65536, False, True, "f16", 1, "llm"),
    ("t1_v16_k65536_r0_bf16_perm", 2048, 2048 + 16, 16, 65536, 0, True, False, "bf16", 1, "llm"),
    # other published members of the family: small / mid residual tables as a second table, fewer main centroids
    ("t1_v16_k65536_r1024_4096x4096", 4096, 4096, 16, 65536, 1024, False, False, "f16", 1, "llm"),
    ("t1_v8_k65536_r4_bias", 2048, 2048, 8, 65536, 4, False, True, "f16", 1, "llm"),
    ("t1_v8_k32768_r0_perm", 2048, 4096, 8, 32768, 0, True, False, "f16", 1, "llm"),
    # 2 - 4 tokens of large-codebook layers: one launch over the sliced layouts, 1 / 4 / 2 / 1 column phases (gemv_sliced_tok.hip)
    ("t2_k65536_r256_4096x4096", 4096, 4096, 8, 65536, 256, False, True, "f16", 2, "llm"),
    ("t4_k65536_r0_8192x2048_perm", 8192, 2048, 8, 65536, 0, True, False, "f16", 4, "llm"),
    ("t3_k65536_r65536_bf16", 4096, 2048 + 8, 8, 65536, 65536, False, True, "bf16", 3, "llm"),
    ("t2_v16_k65536_r65536_4096x2048", 4096, 2048, 16, 65536, 65536, False, False, "f16", 2, "llm"),
]


def main():
    vptq = load_reference()
    torch.set_num_threads(1)  # one summation order, whatever the box
    os.makedirs(os.path.join(HERE, "fmt"), exist_ok=True)
    force = "--force" in sys.argv
    for ci, (name, I, O, v, k, kr, perm, bias, dtype, tokens, dist) in enumerate(FMT_CASES):
        if os.path.exists(os.path.join(HERE, "fmt", f"{name}.npz")) and not force:
            continue   # (fixtures are immutable once committed)
        seed = 5151 + ci
        dt = gg.TORCH_DT[dtype]
        kw = dict(vector_lens=[-1, v], num_centroids=[-1, k], num_res_centroids=[-1, kr if kr > 0 else -1], group_num=1,
                  outlier_size=0, enable_norm=True, enable_perm=perm, bias=bias)
        m = vptq.VQuantLinear(I, O, group_size=I, indices_as_float=False, is_indice_packed=True,
                              dtype=dt, enable_proxy_error=False, **kw)
        t = big_tensors(I, O, v, k, kr, perm, bias, tokens, dtype, dist, seed)

        def f(bits_):  # uint16 bit patterns -> tensor of the layer dtype
            return torch.from_numpy(bits_.view(np.int16).copy()).view(dt)

        assert tuple(m.indices.shape) == tuple(t["indices"].shape), (m.indices.shape, t["indices"].shape)
        m.indices.data = torch.from_numpy(t["indices"].copy()).reshape(m.indices.shape)
        m.centroids.weight.data = f(t["centroids"]).reshape(m.centroids.weight.shape)
        if kr > 0:
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
                   seed=seed, perm=perm, bias=bias, vector_len=v, num_centroids=k,
                   num_res_centroids=kr, W_sha256=hashlib.sha256(Wb.tobytes()).hexdigest(),
                   torch=torch.__version__)
        out = os.path.join(HERE, "fmt", f"{name}.npz")
        np.savez_compressed(out, config=np.frombuffer(json.dumps(cfg).encode(), dtype=np.uint8),
                            y=gg.bits(y), W_head=Wb[:2].copy())
        print(f"{name}: W{tuple(W.shape)} y{tuple(y.shape)} -> {os.path.getsize(out) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
