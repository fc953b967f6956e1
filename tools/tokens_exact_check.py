#!/usr/bin/env python3
"""2 - 4 tokens of the canonical format in the reference's roundings: the persistent MFMA kernel (round 6: the one-token exact loop
with one more pair of MFMAs per token; 3 - 4 tokens of wider layers as two passes of the 2-slot kernel) against the VALU kernel.
Per dtype x shape x token count: the kernel either route names, both outputs against dequant + a float64 product, the largest
difference of the two fp32 outputs (same weights, another summation order), us per launch from a hipGraph.

    python tools/tokens_exact_check.py [--dtypes f16,bf16] [--shapes "8192,8192;4096,4096"] [--tokens 2,3,4]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gpu_gate_count as gc  # noqa: E402
from _gpu_util import gemv_abi, kernel_name  # noqa: E402
from microbench import time_graph  # noqa: E402
from vptq_amd import _backend as B  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtypes", default="f16,bf16")
    ap.add_argument("--shapes", default="8192,8192;4096,4096;4096,14336;14336,4096;2048,8192;8192,1024;11008,4096")
    ap.add_argument("--tokens", default="2,3,4")
    ap.add_argument("--ring", type=int, default=6)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(21)
    routes = (("auto", B.GEMV_EXACT), ("mfma", B.GEMV_EXACT | B.GEMV_FORCE_MFMA), ("valu", B.GEMV_EXACT | B.GEMV_FORCE_VALU))
    for dn in a.dtypes.split(","):
        dt = torch.float16 if dn == "f16" else torch.bfloat16
        for sh in a.shapes.split(";"):
            I, O = (int(v) for v in sh.split(","))
            ring = [gc.make(I, O, "ckpt", dt, dev, g) for _ in range(a.ring)]   # distinct layers: HBM-cold weights
            for T in (int(t) for t in a.tokens.split(",")):
                x = (torch.randn(1, T, I, device=dev, generator=g)).to(dt)
                W = ring[0].dequant()
                ref = x.reshape(T, I).double() @ W.double().t()
                out = {}
                row = []
                for name, fl in routes:
                    y = gemv_abi(ring[0], x, flags=fl, out_f32=True).reshape(T, O).double()
                    out[name] = y
                    err = float((y - ref).abs().max() / ref.abs().max())
                    us = time_graph(lambda fl=fl: [gemv_abi(m, x, flags=fl, workspace=False) for m in ring], 10) / a.ring
                    row.append(f"{name} {kernel_name(ring[0], T, fl)} {us:6.2f} us err {err:.1e}")
                d = float((out["mfma"] - out["valu"]).abs().max() / ref.abs().max())
                print(f"{dn:4s} in {I:5d} out {O:5d} tokens {T}: " + " | ".join(row) + f" | mfma-valu {d:.1e}", flush=True)
            del ring
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
