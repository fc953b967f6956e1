#!/usr/bin/env python3
"""Where does the fused GEMV / batched-decode kernel (calls of <= 16 tokens) stop beating
dequant + dense GEMM?  Per shape and token count, a ring of distinct layers, hipGraph replay.

    python tools/tokens_crossover.py --shapes "4096,4096;8192,8192" --tokens 4,8,12,16,24,32
"""
import argparse, json, os, sys, torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from vptq_amd import _backend as B
from _gpu_util import module_desc
from microbench import time_graph
from shape_bench import mk


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="4096,4096;8192,8192;4096,14336")
    ap.add_argument("--tokens", default="4,8,12,16,24,32,64")
    ap.add_argument("--k", type=int, default=256)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    dev = torch.device("cuda", 0); g = torch.Generator(device=dev).manual_seed(0); lib = B.lib()
    res = []
    for I, O in [tuple(int(v) for v in p.split(',')) for p in a.shapes.split(';')]:
        R = max(2, min(16, (256 << 20) // ((O // 8) * I * 2)))
        layers = [mk(I, O, dev, g) for _ in range(R)]
        descs = [module_desc(m) for m in layers]
        st = B.current_stream_ptr
        for T in [int(t) for t in a.tokens.split(',')]:
            x = torch.randn(1, T, I, device=dev, dtype=torch.float16)
            y = torch.empty(1, T, O, device=dev, dtype=torch.float16)

            def gemv():
                for d, _ in descs:
                    for t0 in range(0, T, 16):   # the library takes <= 16 tokens per call
                        m = min(16, T - t0)
                        B.check(lib.vptq_quant_gemv(d, x[0, t0].data_ptr(), y[0, t0].data_ptr(), m, 0, None, 0,
                                                    torch.cuda.current_stream().cuda_stream), "gemv")

            def gemm():
                for m in layers:
                    F.linear(x, m.dequant())

            tg = time_graph(gemv, 10) / R
            tm = time_graph(gemm, 10) / R
            r = dict(I=I, O=O, tokens=T, gemv_chunks_us=tg, dequant_gemm_us=tm)
            print(json.dumps(r), flush=True); res.append(r)
        del layers, descs
    if a.out:
        json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
