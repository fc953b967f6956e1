#!/usr/bin/env python3
"""2 - 8 tokens of the one-table large-codebook formats in the reference's roundings (VPTQ_GEMV_EXACT): the gather kernel (one
launch for all tokens, entries through the caches) against ONE launch over the exact sliced layout (gemv_sliced_tok.hip, EX) and
against one exact sliced launch per token; ring of distinct layers in a hipGraph, us per layer.
    python tools/sliced_tokens_exact_bench.py --kr 256 --shapes "8192,8192;4096,4096;4096,14336;14336,4096" """
import argparse, json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from vptq_amd import _backend as B  # noqa
from vptq_amd.utils.sliced import SlicedGemv  # noqa
from _gpu_util import module_desc  # noqa
from microbench import time_graph  # noqa
from shape_bench import mk  # noqa

ap = argparse.ArgumentParser()
ap.add_argument("--shapes", default="8192,8192;4096,4096")
ap.add_argument("--ring", type=int, default=8)
ap.add_argument("--k", type=int, default=65536)
ap.add_argument("--kr", type=int, default=256)
ap.add_argument("--v", type=int, default=8)
ap.add_argument("--bf16", action="store_true")
ap.add_argument("--tokens", default="2,3,4,6,8")
a = ap.parse_args()
dev = torch.device("cuda", 0); g = torch.Generator(device=dev).manual_seed(0); lib = B.lib()
dt = torch.bfloat16 if a.bf16 else torch.float16
for I, O in [tuple(int(v) for v in p.split(',')) for p in a.shapes.split(';')]:
    layers = [mk(I, O, dev, g, k=a.k, kr=a.kr, v=a.v) for _ in range(a.ring)]
    if a.bf16:
        layers = [m.to(torch.bfloat16) for m in layers]
    descs = [module_desc(m) for m in layers]
    sls = [SlicedGemv(m, exact=True) for m in layers]
    row = dict(I=I, O=O, v=a.v, k=a.k, kr=a.kr, dtype="bf16" if a.bf16 else "f16", slices=sls[0].slices)
    for T in [int(t) for t in a.tokens.split(',')]:
        x = torch.randn(1, T, I, device=dev).to(dt)
        y = torch.empty(1, T, O, device=dev, dtype=dt)

        def run_gather():
            for d, _ in descs:
                B.check(lib.vptq_quant_gemv(d, x.data_ptr(), y.data_ptr(), T, B.GEMV_EXACT, None, 0, torch.cuda.current_stream().cuda_stream), "gemv")
        B.check(lib.vptq_quant_gemv(descs[0][0], x.data_ptr(), y.data_ptr(), T, B.GEMV_EXACT, None, 0, torch.cuda.current_stream().cuda_stream), "gemv")
        ref = y.clone()
        torch.cuda.synchronize()   # (time_graph works on a stream of its own: its warm-up would overwrite y while the clone still reads it)
        r = dict(gather_us=round(time_graph(run_gather, 10) / a.ring, 2))
        xs = [x[:, t:t + 1].contiguous() for t in range(T)]
        ys = [y[:, t:t + 1] for t in range(T)]
        r["sliced_per_token_us"] = round(time_graph(lambda: [[s(xs[t]) for t in range(T)] for s in sls], 10) / a.ring, 2)
        if all(s.tokens_supported(T) for s in sls):
            got = sls[0].forward_tokens(x)
            torch.cuda.synchronize()
            r["rel_diff"] = ((got.float() - ref.float()).abs().max() / ref.float().abs().max()).item()
            if r["rel_diff"] > 1e-3:    # which of the two is off?  (the gather kernel again, one launch per token)
                B.check(lib.vptq_quant_gemv(descs[0][0], x.data_ptr(), y.data_ptr(), T, B.GEMV_EXACT, None, 0, torch.cuda.current_stream().cuda_stream), "gemv")
                per = torch.cat([sls[0](xs[t]) for t in range(T)], dim=1)
                torch.cuda.synchronize()
                f = lambda a, b: ((a.float() - b.float()).abs().max() / b.float().abs().max()).item()  # noqa: E731
                r["diag"] = dict(gather_again_vs_first=f(y, ref), one_launch_vs_gather_again=f(got, y), per_token_vs_gather_again=f(per, y),
                                 one_launch_again_vs_first=f(sls[0].forward_tokens(x), got))
            r["identical"] = round((got.view(torch.int16) == ref.view(torch.int16)).float().mean().item(), 4)
            r["one_launch_us"] = round(time_graph(lambda: [s.forward_tokens(x) for s in sls], 10) / a.ring, 2)
        row[f"t{T}"] = r
    print(json.dumps(row), flush=True)
    del layers, descs, sls
