#!/usr/bin/env python3
"""Per-shape decode GEMV timing (single launch per layer, ring of distinct layers, hipGraph)
for the projection shapes of Llama-3-8B / 70B, 1..4 tokens."""
import argparse, json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import vptq_amd
from vptq_amd import _backend as B
from _gpu_util import module_desc
from microbench import time_graph

SHAPES = {"8b": [(4096, 4096), (4096, 1024), (4096, 14336), (14336, 4096)],
          "70b": [(8192, 8192), (8192, 1024), (8192, 28672), (28672, 8192)]}


def mk(I, O, dev, g, k=256, kr=256, v=8):
    m = vptq_amd.VQuantLinear(I, O, vector_lens=[-1, v], num_centroids=[-1, k], num_res_centroids=[-1, kr if kr > 0 else -1],
                              group_num=1, group_size=I, outlier_size=0, indices_as_float=False, enable_norm=True,
                              enable_perm=False, is_indice_packed=True, bias=False, dtype=torch.float16, device=dev,
                              enable_proxy_error=False)
    m.indices.data = torch.randint(-2**31, 2**31 - 1, m.indices.shape, generator=g, device=dev, dtype=torch.int64).to(torch.int32)
    m.centroids.weight.data = (torch.randn(m.centroids.weight.shape, generator=g, device=dev) * 0.02).half()
    if kr > 0:
        m.res_centroids.weight.data = (torch.randn(m.res_centroids.weight.shape, generator=g, device=dev) * 0.005).half()
    m.weight_scale.data = (1 + 0.1 * torch.randn(I, generator=g, device=dev)).half()
    m.weight_bias.data = (0.01 * torch.randn(I, generator=g, device=dev)).half()
    return m


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--model", default="70b"); ap.add_argument("--out", default="")
    ap.add_argument("--flags", type=int, default=0, help="VPTQ_GEMV_* flags (1 = fast math)")
    ap.add_argument("--tokens", default="1,2,4")
    ap.add_argument("--shapes", default="", help="I,O;I,O;... instead of --model")
    ap.add_argument("--k", type=int, default=256); ap.add_argument("--kr", type=int, default=256)
    a = ap.parse_args()
    dev = torch.device("cuda", 0); g = torch.Generator(device=dev).manual_seed(0); lib = B.lib()
    res = []
    shapes = ([tuple(int(v) for v in p.split(',')) for p in a.shapes.split(';')] if a.shapes
              else SHAPES[a.model])
    for I, O in shapes:
        import math
        T = int(math.log2(a.k)) + (int(math.log2(a.kr)) if a.kr > 0 else 0)
        idx_bytes = (O // 8) * ((I * T + 31) // 32) * 4
        R = max(2, min(64, (512 << 20) // idx_bytes))
        mods = [mk(I, O, dev, g, a.k, a.kr) for _ in range(R)]
        descs = [module_desc(m) for m in mods]
        for tokens in [int(t) for t in a.tokens.split(',')]:
            x = torch.randn(1, tokens, I, device=dev, dtype=torch.float16)
            ys = [torch.empty(1, tokens, O, device=dev, dtype=torch.float16) for _ in range(R)]
            def run():
                sp = torch.cuda.current_stream().cuda_stream
                for (d, k), y in zip(descs, ys):
                    assert lib.vptq_quant_gemv(d, x.data_ptr(), y.data_ptr(), tokens, a.flags, None, 0, sp) == 0
            us = time_graph(run, 10) / R
            ab = idx_bytes + (a.k + max(a.kr, 0)) * 16 + tokens * 2 * I + 4 * I + tokens * 2 * O
            res.append(dict(I=I, O=O, tokens=tokens, flags=a.flags, kernel=(lib.vptq_quant_gemv_kernel_name(descs[0][0], tokens, a.flags) or b'?').decode(), ring=R, us_per_launch=us, GBps=ab / us / 1e3,
                            frac_hbm=ab / us / 1e3 / 8000))
            print(json.dumps(res[-1]), flush=True)
        del mods, descs
        torch.cuda.empty_cache()
    if a.out:
        json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
