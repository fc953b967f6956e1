#!/usr/bin/env python3
"""Ring of R distinct layers, one decode token per layer, every mode captured in a hipGraph and
replayed interleaved on the same box: us per layer for
  single : one vptq_quant_gemv launch per layer (the library's default kernel)
  t1     : one launch of the chain kernel (gemv_k256c) per layer
  chainN : the ring as launches of N layers each (independent layers)
  groupN : the ring as vptq_quant_gemv_grouped launches of N layers each (the sibling-group route of the module)
  dep    : the ring as ONE dependent chain (x of layer i + 1 is y of layer i)
  a trailing x (singlex, chain32x): the reference's roundings (VPTQ_GEMV_EXACT); chain32s: VPTQ_GEMV_SELECTIVE
--soak S: every mode additionally replayed back to back for S seconds with package power / shader clock sampled
python tools/chain_bench.py --hidden 8192 [--rows O] [--ring 32] [--reps 5] [--libs name=path,...]"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hidden", type=int, default=8192)
    ap.add_argument("--rows", type=int, default=0)
    ap.add_argument("--ring", type=int, default=0)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--modes", default="single,t1,chain4,chain32,dep")
    ap.add_argument("--bf16", action="store_true")
    ap.add_argument("--soak", type=float, default=0.0)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    import bench
    from vptq_amd.ops.chain import GemvChain
    from vptq_amd import _backend as B
    dev = torch.device("cuda", 0)
    I = a.hidden
    O = a.rows or I
    R = a.ring or max(4, (512 << 20) // (O // 8 * I * 2))
    g = torch.Generator(device=dev).manual_seed(1)
    ring = [bench.make_layer(I, O, dev, g) for _ in range(R)]
    if a.bf16:
        for m in ring:
            m.to(torch.bfloat16)
    dt = torch.bfloat16 if a.bf16 else torch.float16
    x = torch.randn(1, 1, I, device=dev, generator=g).to(dt)
    xs = [x] * R
    ys = [torch.empty(1, 1, O, dtype=dt, device=dev) for _ in range(R)]
    alg = bench.alg_bytes(I, O)

    def run_single(fl=0):
        for m, y in zip(ring, ys):
            d = m._descriptor()
            B.check(d[4](d[1], x.data_ptr(), y.data_ptr(), 1, fl, None, 0, B.current_stream_ptr(dev)), "gemv")

    groups = {}

    def run_group(n, fl=0):
        # launches of n layers through vptq_quant_gemv_grouped (what SiblingGroup issues for q/k/v and gate/up)
        import ctypes as C
        if n not in groups:
            gl = []
            for i in range(0, R, n):
                ds = [m._descriptor()[1] for m in ring[i:i + n]]
                k = len(ds)
                gl.append((k, (B.LayerDesc * k)(*ds), (C.c_void_p * k)(*[x.data_ptr()] * k), (C.c_void_p * k)(*[y.data_ptr() for y in ys[i:i + k]])))
            groups[n] = gl
        for k, arr, xp, yp in groups[n]:
            B.check(B.lib().vptq_quant_gemv_grouped(arr, k, xp, yp, 1, fl, B.current_stream_ptr(dev)), "grouped")

    chains = {}

    def run_chain(n, dependent=False, fl=0):
        key = (n, dependent)
        if key not in chains:
            if dependent:
                assert I == O
                chains[key] = [GemvChain(ring, dependent=True)]
            else:
                chains[key] = [GemvChain(ring[i:i + n]) for i in range(0, R, n)]
        if dependent:
            chains[key][0]([x], ys, flags=8 | fl)
        else:
            for i, c in enumerate(chains[key]):
                c(xs[i * n:(i + 1) * n], ys[i * n:(i + 1) * n], flags=8 | fl)

    modes = {}
    for name in a.modes.split(","):
        fl = B.GEMV_EXACT if name.endswith("x") else (B.GEMV_SELECTIVE if name.startswith("chain") and name.endswith("s") else 0)
        base = name[:-1] if fl else name
        if base == "single":
            modes[name] = (lambda fl: (lambda: run_single(fl)))(fl)
        elif name == "t1":
            modes[name] = lambda: run_chain(1)
        elif base.startswith("group"):
            modes[name] = (lambda n, fl: (lambda: run_group(n, fl)))(int(base[5:]), fl)
        elif base.startswith("chain"):
            n = int(base[5:])
            modes[name] = (lambda n, fl: (lambda: run_chain(n, False, fl)))(n, fl)
        elif name == "dep":
            modes[name] = lambda: run_chain(R, True)
    graphs = {}
    s = torch.cuda.Stream()
    ref = None
    parity = {}
    for name, fn in modes.items():
        fn()
        torch.cuda.synchronize()
        if name != "dep":
            out = torch.stack([y.float() for y in ys])
            if ref is None:
                ref = out
            parity[name] = float((out - ref).abs().max() / ref.abs().max())
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.stream(s):
            with torch.cuda.graph(gr, stream=s):
                fn()
        graphs[name] = gr
    res = {k: [] for k in graphs}
    for rep in range(a.reps):
        for name, gr in graphs.items():
            gr.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                gr.replay()
            e1.record()
            torch.cuda.synchronize()
            res[name].append(e0.elapsed_time(e1) * 1e3 / (a.iters * R))
    soak = {}
    if a.soak > 0:
        import time
        for name, gr in graphs.items():
            with bench.SclkSampler(0) as sm:
                t0 = time.time()
                n = 0
                while time.time() - t0 < a.soak:
                    for _ in range(50):
                        gr.replay()
                    n += 50
                    torch.cuda.synchronize()
                dt_s = time.time() - t0
            ps, cs = sm.power_summary(), sm.summary()
            soak[name] = {"us_per_layer": round(dt_s * 1e6 / (n * R), 3), "power_w": ps and ps["median_w"], "sclk_mhz": cs and cs["median_mhz"]}
    summary = {"hidden": I, "rows": O, "ring": R, "alg_bytes": alg, "dtype": str(dt)}
    for name, v in res.items():
        v = sorted(v)
        med = v[len(v) // 2]
        summary[name] = {"us_per_layer": round(med, 3), "min": round(v[0], 3), "max": round(v[-1], 3),
                         "GBps": round(alg / med / 1e3, 1), "frac_8TBps": round(alg / med / 1e3 / 8000, 3),
                         "parity_vs_first": parity.get(name)}
        if name in soak:
            summary[name]["soak"] = soak[name]
        print(f"{name:10s} {med:8.3f} us/layer  ({v[0]:.3f} .. {v[-1]:.3f})  {alg / med / 1e3:8.1f} GB/s  "
              f"frac {alg / med / 1e3 / 8000:.3f}  parity {parity.get(name)}  soak {soak.get(name)}")
    if a.out:
        with open(a.out, "w") as f:
            json.dump(summary, f, indent=1)


if __name__ == "__main__":
    main()
