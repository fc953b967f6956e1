#!/usr/bin/env python3
"""Counting fuzz of the LARGE-CODEBOOK formats (v8-k65536-0 / -256 / -65536) THROUGH THE PRODUCT ROUTE - `VQuantLinear.forward` in
the default arithmetic - on the checkpoint-like tensor families and spiky activations of tools/gpu_gate_count.py, 1 - 3 tokens:
the routes round 5 added for these formats (one token: gemv_sliced<EX>, wide layers in column parts, two tables: RG; 2 / 3 tokens:
one pass, TOK) against W (vptq_dequant: bit-identical to the reference CPU path's dequant) @ x in float64, rounded once.

    python tools/gpu_sliced_count.py [--layers 160] [--dtype f16|bf16] [--seed 0]

Per (format, tokens): layers, which kind of route served them, worst / median of max|dy| / max|y_ref|, count above the bar
(1e-3 fp16, 8e-3 bf16), share of outputs bit-identical to the rounded float64 reference."""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import vptq_amd  # noqa: E402
from gpu_gate_count import make, make_x, XKINDS  # noqa: E402

SHAPES = [(4096, 4096), (8192, 2048), (8192, 4096), (14336, 1024), (28672, 512), (2048, 4096), (4096, 6144), (8192, 1024), (24576, 1024)]
FAMILIES = ("ckpt", "heavy-scale", "outlier-cols", "zero-bias", "t-centroids", "big-bias", "ref-test", "llm-r4")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=160)
    ap.add_argument("--dtype", default="f16", choices=["f16", "bf16"])
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--tokens", type=int, default=3, help="token counts drawn from 1 .. this")
    ap.add_argument("--formats", default="0,256,65536", help="residual codebook sizes drawn (v = 8, 65536 main centroids)")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    dt = torch.float16 if a.dtype == "f16" else torch.bfloat16
    bar = 1e-3 if a.dtype == "f16" else 8e-3
    g = torch.Generator(device=dev).manual_seed(9000 + a.seed)
    rng = np.random.default_rng(a.seed)
    stats = {}
    t0 = time.time()
    # (round 6: VPTQ_ARITHMETIC=selective counts the selective routes - the two-table formats over the folded layouts with the pre-pass)
    assert vptq_amd.arithmetic() in ("reference", "selective")
    krs = tuple(int(v) for v in a.formats.split(","))
    for n in range(a.layers):
        fam = FAMILIES[n % len(FAMILIES)]
        kr = krs[int(rng.integers(0, len(krs)))]
        I, O = SHAPES[int(rng.integers(0, len(SHAPES)))]
        m = make(I, O, fam, dt, dev, g, k=65536, kr=max(kr, 1)) if kr else None
        if m is None:     # (no residual codebook: the module is built without one)
            m = vptq_amd.VQuantLinear(I, O, vector_lens=[-1, 8], num_centroids=[-1, 65536], num_res_centroids=[-1, -1], group_num=1, group_size=I,
                                      outlier_size=0, indices_as_float=False, enable_norm=True, enable_perm=False, is_indice_packed=True,
                                      bias=False, dtype=dt, device=dev, enable_proxy_error=False)
            src = make(I, 8, fam, dt, dev, g, k=65536, kr=4)
            m.centroids.weight.data = src.centroids.weight.data.clone()
            m.weight_scale.data, m.weight_bias.data = src.weight_scale.data.clone(), src.weight_bias.data.clone()
            m.indices.data = torch.randint(-2**31, 2**31 - 1, m.indices.shape, generator=g, device=dev, dtype=torch.int64).to(torch.int32)
            m = m.eval()
        m.enable_sliced_layout()      # (every layer over its layout, small ones too: the kernels are what is counted)
        T = int(rng.integers(1, a.tokens + 1))
        x = torch.cat([make_x(m, XKINDS[(n + t) % len(XKINDS)], dt, dev, g) for t in range(T)], dim=1)
        W = m.dequant()
        r16 = (x.reshape(T, I).double() @ W.double().t()).to(dt)
        y = m(x)
        sl = m.__dict__.get("_sliced", (None, None))[1]
        if sl is None:
            route = "gather"
        elif T == 1:
            route = "selective: pre-pass + folded sliced" if getattr(sl, "selective", False) else "sliced, %d column part(s)%s" % (sl.parts, ", RG" if sl._side16 else "")
        else:
            route = ("one pass" if (m._sliced_one_launch(sl, T) and sl.tokens_one_pass(T)) else "column phases" if m._sliced_one_launch(sl, T)
                     else "launch per token" if T <= m._sliced_token_limit(sl) else "gather")
        den = r16.double().abs().amax(dim=1, keepdim=True).clamp_min(1e-30)
        err = float(((y.reshape(T, O).double() - r16.double()).abs() / den).max())
        ident = float((y.reshape(T, O).view(torch.int16) == r16.view(torch.int16)).float().mean())
        st = stats.setdefault((kr, T, route), dict(err=[], ident=[]))
        st["err"].append(err); st["ident"].append(ident)
        if err > bar:
            print(f"# above the bar: layer {n} {fam} {I}x{O} kr={kr} T={T} {route}: {err:.3e}", flush=True)
        del m, W
        if n % 20 == 19:
            print(f"# {n + 1} layers after {time.time() - t0:.0f} s", flush=True)
    print(f"dtype {a.dtype}, bar {bar:g}, {a.layers} layers, seed {a.seed}, arithmetic of the product route: {vptq_amd.arithmetic()}")
    print(f"{'format':16s} {'tokens':>6s} {'route':32s} {'layers':>6s} {'worst':>9s} {'median':>9s} {'> bar':>6s} {'bit-identical (mean)':>21s}")
    total = 0
    for (kr, T, route), st in sorted(stats.items()):
        e = np.array(st["err"])
        ex = int((e > bar).sum())
        total += ex
        print(f"v8-k65536-{kr:<6d} {T:6d} {route:32s} {len(e):6d} {e.max():9.2e} {np.median(e):9.2e} {ex:6d} {np.mean(st['ident']):21.4f}")
    print(f"exceedances through the product route: {total}")
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main())
