#!/usr/bin/env python3
"""Kernel A/B micro-benchmark for the decode GEMV (GPU box only).

    python tools/microbench.py --hidden 8192 [--ring 32] [--iters 20]

Times, with HIP events on the launch stream, for each arithmetic/kernel variant:
  ring   : R distinct layers (R * packed bytes >= 512 MiB so neither L2 nor the
           256 MiB Infinity Cache can hold them), one launch per layer,
           replayed from a hipGraph  -> us per launch INCLUDING launch gaps
  hot    : the same layer over and over (cache resident)
  group  : the R layers in ceil(R/32) grouped launches
Env knobs read by the library at first use: VPTQ_K256_TAB=0|1, VPTQ_K256_REDUCE=0|1.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vptq_amd  # noqa: E402
from vptq_amd import _backend as B  # noqa: E402


def make_layers(H, R, dev, perm=False, k=256, kr=256, dtype=torch.float16):
    g = torch.Generator(device=dev).manual_seed(1234)
    layers = []
    for _ in range(R):
        m = vptq_amd.VQuantLinear(H, H, vector_lens=[-1, 8], num_centroids=[-1, k],
                                  num_res_centroids=[-1, kr if kr > 0 else -1], group_num=1, group_size=H,
                                  outlier_size=0, indices_as_float=False, enable_norm=True,
                                  enable_perm=perm, is_indice_packed=True, bias=False,
                                  dtype=dtype, device=dev, enable_proxy_error=False)
        m.indices.data = torch.randint(-2**31, 2**31 - 1, m.indices.shape, generator=g,
                                       device=dev, dtype=torch.int64).to(torch.int32)
        m.centroids.weight.data = (torch.randn(m.centroids.weight.shape, generator=g, device=dev) * 0.02).to(dtype)
        if kr > 0:
            m.res_centroids.weight.data = (torch.randn(m.res_centroids.weight.shape, generator=g, device=dev) * 0.005).to(dtype)
        m.weight_scale.data = (1 + 0.1 * torch.randn(H, generator=g, device=dev)).to(dtype)
        m.weight_bias.data = (0.01 * torch.randn(H, generator=g, device=dev)).to(dtype)
        if perm:
            m.perm.data = torch.randperm(H, generator=g, device=dev).to(torch.int32).to(torch.int16)
        if os.environ.get("MB_SHARE_META") and layers:
            # experiment: every layer reads layer 0's codebooks / scale / bias (hot), only the
            # index streams stay distinct (cold): the ceiling of a metadata read-ahead
            m0 = layers[0]
            m.centroids.weight = m0.centroids.weight
            if kr > 0:
                m.res_centroids.weight = m0.res_centroids.weight
            m.weight_scale, m.weight_bias = m0.weight_scale, m0.weight_bias
        layers.append(m)
    return layers


def alg_bytes(H, perm=False, k=256, kr=256):
    import math
    T = int(math.log2(k)) + (int(math.log2(kr)) if kr > 0 else 0)
    return H // 8 * (H * T // 32) * 4 + (k + max(kr, 0)) * 8 * 2 + 2 * H + 4 * H + (2 * H if perm else 0) + 2 * H


def time_graph(fn, iters, warm=3):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            fn()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(iters):
            g.replay()
        e1.record(s)
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters  # us per replay


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hidden", type=int, default=8192)
    ap.add_argument("--ring", type=int, default=0)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--perm", action="store_true")
    ap.add_argument("--group", type=int, default=32)
    ap.add_argument("--prefetch", action="store_true")
    ap.add_argument("--k", type=int, default=256)
    ap.add_argument("--kr", type=int, default=256)
    ap.add_argument("--out", default="")
    ap.add_argument("--variants", default="", help="comma list (default,valu,mfma,exact,exact_mfma,generic); empty = all")
    ap.add_argument("--no-copy", action="store_true", help="skip the 1 GiB copy calibration")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    H = a.hidden
    import math
    Tb = int(math.log2(a.k)) + (int(math.log2(a.kr)) if a.kr > 0 else 0)
    idx_bytes = H // 8 * H * Tb // 8
    R = a.ring or max(2, (512 << 20) // idx_bytes)
    layers = make_layers(H, R, dev, a.perm, a.k, a.kr)
    x = torch.randn(1, 1, H, device=dev, dtype=torch.float16)
    ab = alg_bytes(H, a.perm, a.k, a.kr)
    lib = B.lib()
    st = None
    res = dict(hidden=H, ring=R, alg_bytes=ab, env={k: v for k, v in os.environ.items() if k.startswith("VPTQ_")})

    # calibration: plain device copy of 1 GiB
    if not a.no_copy:
        src = torch.empty(1 << 28, dtype=torch.float32, device=dev); dst = torch.empty_like(src)
        t = time_graph(lambda: dst.copy_(src), 10)
        res["copy_1GiB_TBps"] = 2 * src.numel() * 4 / t / 1e6
        del src, dst

    from tests_gpu_util import module_desc  # noqa
    descs, keeps, ys = [], [], []
    for i, m in enumerate(layers):
        d, k = module_desc(m, prefetch=layers[(i + 1) % R].indices if a.prefetch else None)
        descs.append(d); keeps.append(k); ys.append(torch.empty(1, 1, H, device=dev, dtype=torch.float16))

    def launch_one(i, flags):
        rc = lib.vptq_quant_gemv(descs[i], x.data_ptr(), ys[i].data_ptr(), 1, flags, None, 0,
                                 torch.cuda.current_stream().cuda_stream)
        assert rc == 0, lib.vptq_last_error()

    chunks = []
    for i0 in range(0, R, a.group):
        m = min(a.group, R - i0)
        chunks.append((m, (B.LayerDesc * m)(*descs[i0:i0 + m]), (C.c_void_p * m)(*[x.data_ptr()] * m),
                       (C.c_void_p * m)(*[y.data_ptr() for y in ys[i0:i0 + m]])))

    def launch_group(flags):
        for m, arr, xp, yp in chunks:
            rc = lib.vptq_quant_gemv_grouped(arr, m, xp, yp, 1, flags,
                                             torch.cuda.current_stream().cuda_stream)
            assert rc == 0, lib.vptq_last_error()

    # VPTQ_GEMV_* flags: 0 = default (folded arithmetic, kernel chosen by launch size),
    # 4 = EXACT, 8 = FORCE_MFMA, 16 = FORCE_VALU, 2 = FORCE_GENERIC
    variants = ((("default", 0), ("valu", 16), ("mfma", 8), ("exact", 4), ("exact_mfma", 12), ("generic", 2))
                if a.k == 256 else (("default", 0), ("generic", 2)))
    if a.variants:
        variants = tuple(v for v in variants if v[0] in a.variants.split(","))
    for name, flags in variants:
        ring = time_graph(lambda: [launch_one(i, flags) for i in range(R)], a.iters) / R
        hot = time_graph(lambda: [launch_one(0, flags) for _ in range(R)], a.iters) / R
        r = dict(ring_us=ring, ring_TBps=ab / ring / 1e6, hot_us=hot, hot_TBps=ab / hot / 1e6)
        if flags != 2:
            grp = time_graph(lambda: launch_group(flags), a.iters) / R
            r.update(group_us=grp, group_TBps=ab / grp / 1e6)
        res[name] = r
        print(name, json.dumps(r), flush=True)
    print(json.dumps(res))
    if a.out:
        os.makedirs(os.path.dirname(a.out), exist_ok=True)
        with open(a.out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _gpu_util
    sys.modules["tests_gpu_util"] = _gpu_util
    main()
