#!/usr/bin/env python3
"""`VQuantLinear.forward`, one token, per index format x dtype x arithmetic: a ring of distinct 8192^2 layers (HBM-cold) replayed
from a hipGraph, us per layer and the kernel the module's route names.

    python tools/module_route_bench.py [--hidden 8192] [--formats k256-256,k65536-0,k65536-256,k65536-65536] [--dtypes f16,bf16]
                                       [--arithmetics reference,selective]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench  # noqa: E402
import vptq_amd  # noqa: E402
from microbench import time_graph  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hidden", type=int, default=8192)
    ap.add_argument("--formats", default="k256-256,k65536-0,k65536-256,k65536-65536")
    ap.add_argument("--dtypes", default="f16,bf16")
    ap.add_argument("--arithmetics", default="reference,selective")
    ap.add_argument("--ring", type=int, default=8)
    ap.add_argument("--tokens", type=int, default=1)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    H = a.hidden
    for fmt in a.formats.split(","):
        k, kr = (int(t) for t in fmt[1:].split("-"))
        for dn in a.dtypes.split(","):
            dt = torch.float16 if dn == "f16" else torch.bfloat16
            g = torch.Generator(device=dev).manual_seed(3)
            for ar in a.arithmetics.split(","):
                vptq_amd.set_arithmetic(ar)
                ring = [bench.make_layer(H, H, dev, g, k, kr, dtype=dt) for _ in range(a.ring)]
                x = torch.randn(1, a.tokens, H, device=dev, generator=g).to(dt)
                with torch.no_grad():
                    for m in ring:
                        m(x)
                    us = time_graph(lambda: [m(x) for m in ring], 20) / a.ring
                route = getattr(ring[0], "last_route", None)
                print(f"v8-{fmt:12s} {dn:4s} {ar:10s} tokens {a.tokens}: {us:7.2f} us per layer   {route() if callable(route) else route or ''}", flush=True)
                del ring
                torch.cuda.empty_cache()
    vptq_amd.set_arithmetic("reference")


if __name__ == "__main__":
    main()
