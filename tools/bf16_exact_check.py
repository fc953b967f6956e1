#!/usr/bin/env python3
"""bf16, the reference's roundings (default arithmetic), one token: the matrix-pipe form inside gemv_k256m_kernel
(round 6, vptq_amd/csrc/gemv_k256m.hip sweep(), kSB) against the widened VALU kernel (gemv_k256_kernel) it replaces from 32
row groups on.  Per shape: the kernel each route names, the largest difference of the two fp32 outputs (they round every
weight the same way - only the fp32 summation order differs), both against dequant (bit-identical to the reference's CPU
dequant: tests/test_hip_parity.py) + a float64 product, and the time of either launch.

    python tools/bf16_exact_check.py [--reps 200]"""
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
from vptq_amd import _backend as B  # noqa: E402


def timed(fn, reps):
    # one launch per replay step of a hipGraph holding 8 of them: the kernel and its launch gap, no Python
    from microbench import time_graph
    return time_graph(lambda: [fn() for _ in range(8)], max(5, reps // 8)) / 8


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=200)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16"], help="f16: the wide layers (6 - 7 sweeps) whose scale and bias are staged in LDS too")
    ap.add_argument("--shapes", default="", help="in,out;in,out;...")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(11)
    shapes = [(8192, 8192), (4096, 4096), (11008, 4096), (4096, 11008), (12288, 4096), (2048, 8192), (5120, 5120), (1024, 1024)]
    if a.shapes:
        shapes = [tuple(int(v) for v in t.split(",")) for t in a.shapes.split(";")]
    for dt in (torch.bfloat16 if a.dtype == "bf16" else torch.float16,):
        for fam in ("ckpt", "ref-test", "llm-r4"):
            for (I, O) in shapes:
                m = gc.make(I, O, fam, dt, dev, g)
                worst = {}
                names = {}
                for xk in gc.XKINDS:
                    x = gc.make_x(m, xk, dt, dev, g).reshape(1, I)
                    W = m.dequant()
                    ref = W.double() @ x.reshape(-1).double()
                    ys = {}
                    for name, fl in (("mfma", B.GEMV_EXACT), ("valu", B.GEMV_EXACT | B.GEMV_FORCE_VALU)):
                        names[name] = kernel_name(m, 1, fl)
                        ys[name] = gemv_abi(m, x, flags=fl, out_f32=True).reshape(-1).double()
                        worst[name] = max(worst.get(name, 0.0), float((ys[name] - ref).abs().max() / ref.abs().max()))
                    worst["mfma-valu"] = max(worst.get("mfma-valu", 0.0),
                                             float((ys["mfma"] - ys["valu"]).abs().max() / ref.abs().max()))
                    # the product's own output type: how many outputs round differently
                    yb = {n: gemv_abi(m, x, flags=fl).reshape(-1) for n, fl in (("mfma", B.GEMV_EXACT), ("valu", B.GEMV_EXACT | B.GEMV_FORCE_VALU))}
                    worst["bf16 outputs differing"] = max(worst.get("bf16 outputs differing", 0), int((yb["mfma"] != yb["valu"]).sum()))
                x = gc.make_x(m, gc.XKINDS[0], dt, dev, g).reshape(1, I)
                t = {n: timed(lambda fl=fl: gemv_abi(m, x, flags=fl, workspace=False), a.reps)
                     for n, fl in (("mfma", B.GEMV_EXACT), ("valu", B.GEMV_EXACT | B.GEMV_FORCE_VALU))}
                print(f"{fam:9s} in {I:5d} out {O:5d}  {names['mfma']} {t['mfma']:6.2f} us | {names['valu']} {t['valu']:6.2f} us | "
                      + "  ".join(f"{k} {v:.2e}" if isinstance(v, float) else f"{k} {v}" for k, v in worst.items()), flush=True)
                del m
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
