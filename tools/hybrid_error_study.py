#!/usr/bin/env python3
"""CPU study (round 6): the folded form with an EXACT correction on the activation columns that dominate.

    python tools/hybrid_error_study.py [--layers 400] [--seed 0] [--kappa 3,4,6] [--dtype f16]

Same tensor families and activation kinds as tools/gpu_gate_count.py.  Per layer, against the reference (every weight
rounded three times, float64 product rounded once - what the oracle pins on the goldens):
  folded : y = sum (c + r) f16(s x) + sum b x
  hybrid : columns with |f16(s x)| >= kappa * rms(f16(s x)) take the reference's roundings (w = f16(f16(f16(c+r) s) + b), w x),
           the rest the folded form; a UNIT of the kernel (16 columns: one of each 8-column chunk of a 128-column block) is exact
           as soon as one of its columns is
Reports max|dy| / max|y_ref| (worst / median / count above the bar) per activation kind and the share of exact units."""
import argparse
import sys
import time

import numpy as np
import torch

FAMILIES = ("ckpt", "heavy-scale", "outlier-cols", "zero-bias", "t-centroids", "sorted-idx", "low-entropy", "big-bias", "ref-test", "llm-r4")
XKINDS = ("normal", "massive", "relu", "sparse", "bias-orth")
SHAPES = [(4096, 4096), (4096, 1024), (2048, 8192), (1024, 4096), (4096, 2048)]   # (in, out); in = a multiple of 512


def make(I, O, fam, dt, g):
    rn = lambda *s: torch.randn(*s, generator=g)  # noqa: E731
    k = kr = 256
    if fam == "t-centroids":
        t3 = lambda *s: rn(*s) / torch.sqrt((rn(3, *s) ** 2).mean(0))  # noqa: E731
        c, r = t3(k, 8), 0.25 * t3(kr, 8)
    else:
        c, r = rn(k, 8), 0.25 * rn(kr, 8)
    sig = {"heavy-scale": 1.0}.get(fam, 0.3)
    scale = 0.02 * torch.exp(sig * rn(I))
    if fam == "outlier-cols":
        hot = torch.randperm(I, generator=g)[: max(1, I // 200)]
        scale[hot] *= 10 + 40 * torch.rand(hot.numel(), generator=g)
    bias = 0.002 * rn(I)
    if fam == "zero-bias":
        bias = bias * (0.0 if int(torch.randint(0, 2, (1,), generator=g)) else 5e-3)
    if fam == "big-bias":
        bias = 0.05 * rn(I)
    if fam == "ref-test":
        c, r, scale, bias = 0.02 + 0.5 * rn(k, 8), 0.02 + 0.5 * rn(kr, 8), 0.02 + 0.5 * rn(I), 0.02 + 0.5 * rn(I)
    if fam == "llm-r4":
        c, r, scale, bias = 0.02 * rn(k, 8), 0.005 * rn(kr, 8), 1 + 0.1 * rn(I), 0.01 * rn(I)
    N = O // 8
    hi = 16 if fam == "low-entropy" else k
    e = torch.randint(0, hi, (N, I), generator=g)
    er = torch.randint(0, kr, (N, I), generator=g)
    if fam == "sorted-idx":
        e = torch.sort(e, dim=1).values
    return c.to(dt), r.to(dt), scale.to(dt), bias.to(dt), e, er


def make_x(I, kind, bias, dt, g):
    x = torch.randn(I, generator=g)
    if kind == "massive":
        hot = torch.randperm(I, generator=g)[: max(1, I // 200)]
        x[hot] *= 50
    elif kind == "relu":
        x = x.abs()
    elif kind == "sparse":
        x = x * (torch.rand(I, generator=g) < 0.1)
    elif kind == "bias-orth":
        b = bias.float()
        bb = (b * b).sum()
        if float(bb) > 0:
            x = x - (x * b).sum() / bb * b
    return x.to(dt)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=400)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--kappa", default="3,4,6")
    ap.add_argument("--dtype", default="f16")
    ap.add_argument("--threads", type=int, default=2)
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    dt = torch.float16 if a.dtype == "f16" else torch.bfloat16
    bar = 1e-3 if a.dtype == "f16" else 8e-3
    kappas = [float(s) for s in a.kappa.split(",")]
    g = torch.Generator().manual_seed(9000 + a.seed)
    rng = np.random.default_rng(a.seed)
    names = ["folded"] + [f"hyb{k:g}" for k in kappas] + [f"blk{k:g}" for k in kappas] + [f"loc{k:g}" for k in kappas]
    err = {(xk, n): [] for xk in XKINDS for n in names}
    units = {(xk, n): [] for xk in XKINDS for n in names}
    t0 = time.time()
    for i in range(a.layers):
        fam = FAMILIES[i % len(FAMILIES)]
        xk = XKINDS[(i // len(FAMILIES)) % len(XKINDS)]
        I, O = SHAPES[int(rng.integers(0, len(SHAPES)))]
        c, r, s, b, e, er = make(I, O, fam, dt, g)
        x = make_x(I, xk, b, dt, g)
        # [N, I, 8] -> [O, I]
        cr16 = (c[e] + r[er])                                    # rounded to the 16-bit type (torch rounds each op)
        W = ((cr16 * s[None, :, None]) + b[None, :, None])       # two more roundings
        W = W.permute(0, 2, 1).reshape(O, I)
        A = (c.double()[e] + r.double()[er]).permute(0, 2, 1).reshape(O, I)   # un-rounded c + r
        xd = x.double()
        y_ref = (W.double() @ xd).to(dt).double()
        den = y_ref.abs().max().clamp_min(1e-30)
        xs = (s * x)                                              # f16(s x)
        xsd = xs.double()
        bx = b.double() * xd
        y_f = (A @ xsd + bx.sum()).float().to(dt).double()
        err[(xk, "folded")].append(float((y_f - y_ref).abs().max() / den))
        rms = float(xsd.pow(2).mean().sqrt())
        for k in kappas:
            hot = xsd.abs() >= k * rms
            # a unit = columns {blk * 128 + chunk * 8 + u}: exact as soon as one of them is hot
            hu = hot.view(-1, 16, 8).any(dim=1)                  # [blocks, u]
            ex = hu[:, None, :].expand(-1, 16, -1).reshape(-1)
            units[(xk, f"hyb{k:g}")].append(float(hu.float().mean()))
            y_h = (A[:, ~ex] @ xsd[~ex] + bx[~ex].sum() + W[:, ex].double() @ xd[ex]).float().to(dt).double()
            err[(xk, f"hyb{k:g}")].append(float((y_h - y_ref).abs().max() / den))
            # what the kernels do (round 6): blocks of 128 consecutive columns.  blk: threshold from the layer's rms
            # (gemv_k256c: k256c_hot_kernel); loc: from the rms of the 512 columns a wave stages (gemv_k256m: no barrier needed)
            for nm, hotc in ((f"blk{k:g}", hot), (f"loc{k:g}", xsd.abs() >= k * xsd.view(-1, 512).pow(2).mean(dim=1).sqrt().repeat_interleave(512))):
                hb = hotc.view(-1, 128).any(dim=1)
                ex = hb.repeat_interleave(128)
                units[(xk, nm)].append(float(hb.float().mean()))
                y_h = (A[:, ~ex] @ xsd[~ex] + bx[~ex].sum() + W[:, ex].double() @ xd[ex]).float().to(dt).double()
                err[(xk, nm)].append(float((y_h - y_ref).abs().max() / den))
        if (i + 1) % 50 == 0:
            print(f"# {i + 1} layers, {time.time() - t0:.0f} s", file=sys.stderr, flush=True)
    print(f"dtype {a.dtype} bar {bar:g} layers {a.layers} seed {a.seed}")
    print(f"{'x kind':10s} {'form':8s} {'n':>5s} {'worst':>9s} {'p99':>9s} {'median':>9s} {'> bar':>6s} {'exact units':>12s}")
    for xk in XKINDS:
        for n in names:
            v = np.array(err[(xk, n)])
            if not len(v):
                continue
            u = ""
            if n != "folded":
                u = f"{np.mean(units[(xk, n)]):12.4f}"
            print(f"{xk:10s} {n:8s} {len(v):5d} {v.max():9.2e} {np.quantile(v, 0.99):9.2e} {np.median(v):9.2e} {int((v > bar).sum()):6d} {u}")
    tot = {n: int(sum((np.array(err[(xk, n)]) > bar).sum() for xk in XKINDS)) for n in names}
    print("exceedances:", tot)


if __name__ == "__main__":
    main()
