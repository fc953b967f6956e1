#!/usr/bin/env python3
"""Same-box, same-process A/B of several builds of libvptq_hip.so (GPU box only).

    python tools/ab_libs.py --libs base=vptq_amd/libvptq_hip.so,wpx=tools/_build/libvptq_hip_wpx.so \
        [--hidden 8192] [--reps 3] [--group 4] [--flags 0] [--out gpurun_out/ab.json]

All builds are loaded into ONE process (ctypes), the layers are made once, every build gets its own
captured hipGraphs (ring of distinct layers / the same layer over and over / grouped launches), and
the replays are interleaved rep by rep, so that box-to-box and minute-to-minute drift hits every
build alike.  The first build's output is the reference of a parity check for the others, and
every build is checked against HIP dequant + fp32 matmul (<= 1e-3 of max |y|).
"""
import argparse
import ctypes as C
import json
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import vptq_amd  # noqa: E402,F401
from vptq_amd import _backend as B  # noqa: E402
import _gpu_util  # noqa: E402
sys.modules["tests_gpu_util"] = _gpu_util
import microbench  # noqa: E402


def load(path):
    l = C.CDLL(os.path.abspath(path))
    for name, (res, args) in B.EXPORTS.items():
        fn = getattr(l, name)
        fn.restype, fn.argtypes = res, args
    assert l.vptq_abi_version() == B.ABI_VERSION
    return l


def capture(fn, stream):
    with torch.cuda.stream(stream):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            fn()
        g.replay()
        torch.cuda.synchronize()
    return g


def replay_us(g, stream, iters):
    with torch.cuda.stream(stream):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(iters):
            g.replay()
        e1.record(stream)
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", required=True, help="name=path,name=path,...")
    ap.add_argument("--hidden", type=int, default=8192)
    ap.add_argument("--out-features", type=int, default=0)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--group", type=int, default=4)
    ap.add_argument("--flags", type=int, default=0)
    ap.add_argument("--modes", default="ring,hot,group")
    ap.add_argument("--out", default="")
    ap.add_argument("--dtype", default="f16", choices=["f16", "bf16"])
    ap.add_argument("--no-parity", action="store_true", help="timing-only (ablation) builds: report, do not assert")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    H = a.hidden
    # name=path or name=path@flags (VPTQ_GEMV_* flags of that entry, e.g. 8 = FORCE_MFMA, 16 = FORCE_VALU)
    libs, lib_flags = [], {}
    for kv in a.libs.split(","):
        name, path = kv.split("=")
        path, _, fl = path.partition("@")
        libs.append((name, load(path)))
        lib_flags[name] = int(fl) if fl else a.flags
    idx_bytes = H // 8 * H * 2
    R = max(2, (512 << 20) // idx_bytes)
    R = (R // a.group) * a.group
    tdt = torch.float16 if a.dtype == "f16" else torch.bfloat16
    tol = 1e-3 if a.dtype == "f16" else 8e-3
    layers = microbench.make_layers(H, R, dev, dtype=tdt)
    x = torch.randn(1, 1, H, device=dev, dtype=tdt)
    ab = microbench.alg_bytes(H)
    descs, keeps = [], []
    for m in layers:
        d, k = _gpu_util.module_desc(m)
        descs.append(d); keeps.append(k)
    stream = torch.cuda.Stream()
    modes = a.modes.split(",")

    # parity: every build against HIP dequant + fp32 matmul on layer 0, and against the first build
    W = torch.empty(H, H, device=dev, dtype=tdt)
    d0inv, k0 = _gpu_util.module_desc(layers[0], need_inv_perm=True)
    assert libs[0][1].vptq_dequant(d0inv, W.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
    ref = (x.float().reshape(1, H) @ W.float().t()).reshape(-1)
    outs = {}
    for name, l in libs:
        y = torch.zeros(1, 1, H, device=dev, dtype=tdt)
        rc = l.vptq_quant_gemv(descs[0], x.data_ptr(), y.data_ptr(), 1, lib_flags[name], None, 0,
                               torch.cuda.current_stream().cuda_stream)
        assert rc == 0, l.vptq_last_error()
        torch.cuda.synchronize()
        outs[name] = y.float().reshape(-1)
        err = float((outs[name] - ref).abs().max() / ref.abs().max())
        d0 = float((outs[name] - outs[libs[0][0]]).abs().max() / ref.abs().max())
        kn = l.vptq_quant_gemv_kernel_name(descs[0], 1, lib_flags[name])
        print(f"parity {name:12s} kernel={kn.decode() if kn else None} rel_err_vs_dequant={err:.2e} "
              f"vs_{libs[0][0]}={d0:.2e}", flush=True)
        assert a.no_parity or err <= tol, (name, err)

    graphs = {}
    ys = [torch.empty(1, 1, H, device=dev, dtype=tdt) for _ in range(R)]
    for name, l in libs:
        fl = lib_flags[name]

        def one(i, l=l, fl=fl):
            rc = l.vptq_quant_gemv(descs[i], x.data_ptr(), ys[i].data_ptr(), 1, fl, None, 0,
                                   torch.cuda.current_stream().cuda_stream)
            assert rc == 0, l.vptq_last_error()
        chunks = []
        for i0 in range(0, R, a.group):
            m = a.group
            chunks.append((m, (B.LayerDesc * m)(*descs[i0:i0 + m]), (C.c_void_p * m)(*[x.data_ptr()] * m),
                           (C.c_void_p * m)(*[y.data_ptr() for y in ys[i0:i0 + m]])))

        def grouped(l=l, chunks=chunks, fl=fl):
            for m, arr, xp, yp in chunks:
                rc = l.vptq_quant_gemv_grouped(arr, m, xp, yp, 1, fl,
                                               torch.cuda.current_stream().cuda_stream)
                assert rc == 0, l.vptq_last_error()
        keeps.append(chunks)
        if "ring" in modes:
            graphs[(name, "ring")] = capture(lambda: [one(i) for i in range(R)], stream)
        if "hot" in modes:
            graphs[(name, "hot")] = capture(lambda: [one(0) for _ in range(R)], stream)
        if "group" in modes:
            graphs[(name, "group")] = capture(grouped, stream)

    res = {name: {m: [] for m in modes} for name, _ in libs}
    for rep in range(a.reps):
        for name, _ in libs:
            for m in modes:
                res[name][m].append(replay_us(graphs[(name, m)], stream, a.iters) / R)
    print(f"hidden {H}, ring {R}, {a.reps} interleaved reps x {a.iters} replays; us per layer (median [min..max])")
    summary = {}
    for name, _ in libs:
        row = []
        summary[name] = {}
        for m in modes:
            v = res[name][m]
            med = statistics.median(v)
            summary[name][m] = dict(median_us=med, min_us=min(v), max_us=max(v), TBps=ab / med / 1e6)
            row.append(f"{m} {med:6.2f} [{min(v):5.2f}..{max(v):5.2f}] {ab / med / 1e6:5.2f} TB/s")
        print(f"{name:14s} " + " | ".join(row), flush=True)
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        with open(a.out, "w") as f:
            json.dump(dict(hidden=H, ring=R, flags=a.flags, alg_bytes=ab, raw=res, summary=summary), f, indent=1)


if __name__ == "__main__":
    main()
