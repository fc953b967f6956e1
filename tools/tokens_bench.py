#!/usr/bin/env python3
"""us per layer for 1..16 tokens through vptq_quant_gemv (ring of distinct layers, hipGraph).
VPTQ_GEMM_MIN_TOKENS=99 in the environment = the round-1 path (launches of <= 4 tokens); --no-ws = no scratch
buffer = the round-2 kernels (gemm_k256 for 5-16 tokens) instead of the one-pass gemm_k256t.
    python tools/tokens_bench.py --shapes "8192,8192;4096,4096" --tokens 1,2,4,5,8,16"""
import argparse, json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from vptq_amd import _backend as B
from _gpu_util import module_desc
from microbench import time_graph
from shape_bench import mk

ap = argparse.ArgumentParser()
ap.add_argument("--shapes", default="8192,8192;4096,4096;8192,28672")
ap.add_argument("--tokens", default="1,2,4,5,8,12,16")
ap.add_argument("--out", default="")
ap.add_argument("--no-ws", action="store_true")
ap.add_argument("--bf16", action="store_true")
a = ap.parse_args()
dt = torch.bfloat16 if a.bf16 else torch.float16
dev = torch.device("cuda", 0); g = torch.Generator(device=dev).manual_seed(0); lib = B.lib()
res = []
for I, O in [tuple(int(v) for v in p.split(',')) for p in a.shapes.split(';')]:
    R = max(2, min(32, (512 << 20) // ((O // 8) * I * 2)))
    layers = [mk(I, O, dev, g) for _ in range(R)]
    if a.bf16:
        layers = [m.to(torch.bfloat16) for m in layers]
    descs = [module_desc(m) for m in layers]
    for T in [int(t) for t in a.tokens.split(',')]:
        x = torch.randn(1, T, I, device=dev, dtype=dt)
        y = torch.empty(1, T, O, device=dev, dtype=dt)
        nb = 0 if a.no_ws else lib.vptq_quant_gemv_workspace_bytes(descs[0][0], T, 0)
        ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=dev)

        def run():
            for d, _ in descs:
                B.check(lib.vptq_quant_gemv(d, x.data_ptr(), y.data_ptr(), T, 0, ws.data_ptr() if nb else None, nb,
                                            torch.cuda.current_stream().cuda_stream), "gemv")
        us = time_graph(run, 10) / R
        name = lib.vptq_quant_gemv_kernel_name(descs[0][0], T, 0).decode() + (" (no workspace: the round-2 kernels)" if a.no_ws else "")
        r = dict(I=I, O=O, tokens=T, us=us, kernel=name, dtype="bf16" if a.bf16 else "f16", min_tokens=os.environ.get("VPTQ_GEMM_MIN_TOKENS", "5"))
        print(json.dumps(r), flush=True); res.append(r)
    del layers, descs
if a.out:
    json.dump(res, open(a.out, "w"), indent=1)
