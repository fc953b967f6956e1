#!/usr/bin/env python3
"""Decode GEMV timing per index FORMAT (GPU box only): vector length, codebook sizes, codebook groups.

    python tools/format_bench.py [--hidden 8192] [--formats v16-k65536-65536,v8-k32768-0,...] [--tokens 1]

For every format a ring of distinct layers (>= 512 MiB of packed indices, capped at 32 layers) is launched
one layer per launch from a hipGraph, with the library's kernel choice and with VPTQ_GEMV_FORCE_GENERIC.
Prints one JSON line per format: kernel, us per launch, GB/s of the algorithmic bytes."""
import argparse
import json
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import vptq_amd  # noqa: E402
from vptq_amd import _backend as B  # noqa: E402
from _gpu_util import module_desc  # noqa: E402
from microbench import time_graph  # noqa: E402


def make(H, O, v, k, kr, C, dev, g, dtype=torch.float16, S=0, ov=4, ko=4096):
    m = vptq_amd.VQuantLinear(H, O, vector_lens=[ov if S else -1, v], num_centroids=[ko if S else -1, k],
                              num_res_centroids=[-1, kr if kr > 0 else -1], group_num=C, group_size=(H - S) // C,
                              outlier_size=S, indices_as_float=False, enable_norm=True, enable_perm=False,
                              is_indice_packed=True, bias=False, dtype=dtype, device=dev,
                              enable_proxy_error=False)
    m.indices.data = torch.randint(-2**31, 2**31 - 1, m.indices.shape, generator=g, device=dev,
                                   dtype=torch.int64).to(torch.int32)
    m.centroids.weight.data = (torch.randn(m.centroids.weight.shape, generator=g, device=dev) * 0.02).to(dtype)
    if kr > 0:
        m.res_centroids.weight.data = (torch.randn(m.res_centroids.weight.shape, generator=g, device=dev) * 0.005).to(dtype)
    if S:
        m.outlier_centroids.weight.data = (torch.randn(m.outlier_centroids.weight.shape, generator=g, device=dev) * 0.05).to(dtype)
        m.outlier_indices.data = torch.randint(0, ko, m.outlier_indices.shape, generator=g, device=dev,
                                               dtype=torch.int64).to(torch.int16)   # uint16 bit patterns
    m.weight_scale.data = (1 + 0.1 * torch.randn(H, generator=g, device=dev)).to(dtype)
    m.weight_bias.data = (0.01 * torch.randn(H, generator=g, device=dev)).to(dtype)
    return m


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hidden", type=int, default=8192)
    ap.add_argument("--formats", default="v16-k65536-65536,v16-k65536-32768,v16-k65536-1024,v16-k65536-0,v8-k65536-1024,"
                                         "v8-k32768-0,v8-k16384-16384,v8-k65536-256,v8-k4096-4096-c2,v12-k65536-4096")
    ap.add_argument("--tokens", type=int, default=1)
    ap.add_argument("--dtype", default="f16", choices=["f16", "bf16"])
    ap.add_argument("--dense", action="store_true", help="also time vptq_dequant + F.linear at this token count")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    lib = B.lib()
    H = a.hidden
    tdt = torch.float16 if a.dtype == "f16" else torch.bfloat16
    res = []
    for f in a.formats.split(","):
        parts = f.split("-")
        v, k, kr = int(parts[0][1:]), int(parts[1][1:]), int(parts[2])
        # optional suffixes: -c<codebook groups>, -o<outlier columns> (vector length 4, 4096 entries), -ov<their vector length>
        C = S = 0
        ov = 4
        for q in parts[3:]:
            if q.startswith("c"): C = int(q[1:])
            elif q.startswith("ov"): ov = int(q[2:])
            elif q.startswith("o"): S = int(q[1:])
        C = C or 1
        T = int(math.log2(k)) + (int(math.log2(kr)) if kr > 0 else 0)
        idx_bytes = (H // v) * C * (((H - S) // C * T + 31) // 32) * 4 + (H // ov) * S * 2
        R = max(2, min(32, (512 << 20) // idx_bytes))
        g = torch.Generator(device=dev).manual_seed(0)
        mods = [make(H, H, v, k, kr, C, dev, g, tdt, S, ov) for _ in range(R)]
        descs = [module_desc(m) for m in mods]
        x = torch.randn(1, a.tokens, H, device=dev, dtype=tdt)
        ys = [torch.empty(1, a.tokens, H, device=dev, dtype=tdt) for _ in range(R)]
        ab = idx_bytes + C * (k + max(kr, 0)) * v * 2 + (4096 * ov * 2 if S else 0) + a.tokens * 2 * H + 4 * H + a.tokens * 2 * H
        row = dict(format=f, hidden=H, tokens=a.tokens, dtype=a.dtype, T=T, ring=R, alg_bytes=ab)
        for name, flags in (("default", 0), ("generic", B.GEMV_FORCE_GENERIC)):
            def run(flags=flags):
                sp = torch.cuda.current_stream().cuda_stream
                for (d, kp), y in zip(descs, ys):
                    assert lib.vptq_quant_gemv(d, x.data_ptr(), y.data_ptr(), a.tokens, flags, None, 0, sp) == 0, \
                        lib.vptq_last_error()
            us = time_graph(run, 5) / R
            kn = lib.vptq_quant_gemv_kernel_name(descs[0][0], a.tokens, flags)
            row[name] = dict(kernel=kn.decode() if kn else None, us_per_launch=us, GBps=ab / us / 1e3)
        if a.dense:
            # the other route of ops.quant_gemm: dense W by vptq_dequant, then F.linear (hipBLASLt)
            import torch.nn.functional as F
            W = torch.empty(H, H, device=dev, dtype=tdt)
            descs_inv = [module_desc(m, need_inv_perm=True) for m in mods]

            def dense():
                sp = torch.cuda.current_stream().cuda_stream
                for (d, kp), y in zip(descs_inv, ys):
                    assert lib.vptq_dequant(d, W.data_ptr(), sp) == 0, lib.vptq_last_error()
                    y.copy_(F.linear(x, W))
            us = time_graph(dense, 5) / R
            row["dense"] = dict(kernel="dequant + F.linear", us_per_launch=us)
        y0 = ys[0].clone()
        run(0)
        torch.cuda.synchronize()
        row["max_rel_diff_default_vs_generic"] = float((ys[0].float() - y0.float()).abs().max() / y0.float().abs().max())
        res.append(row)
        print(json.dumps(row), flush=True)
        del mods, descs, ys
        torch.cuda.empty_cache()
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
