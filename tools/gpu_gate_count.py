#!/usr/bin/env python3
"""Counting fuzz of the measured folded-form gate (vptq_amd/_backend.py:folded_form_is_safe, round 5) THROUGH THE PRODUCT
ROUTE - `VQuantLinear.forward` - on tensor families the gate was not tuned on: checkpoint-like
statistics instead of the four synthetic distributions of tools/gpu_fuzz_count.py.

    python tools/gpu_gate_count.py [--layers 4096] [--dtype f16|bf16] [--seed 0]

Families (weights as VPTQ stores them: centroids of column-NORMALISED weights, weight_scale = column std, weight_bias =
column mean - vptq/layers/vqlinear.py:214-240 of the reference):
  ckpt          centroids N(0, 1), residual N(0, 0.25), scale 0.02 lognormal(0.3), bias N(0, 0.002)
  heavy-scale   ckpt with scale 0.02 lognormal(1.0) (heavy-tailed column norms)
  outlier-cols  ckpt with 0.5 % of the columns' scale x 10 ... 50 (outlier channels)
  zero-bias     ckpt with bias exactly 0 / 1e-5 N
  t-centroids   centroids / residual Student-t (3 degrees of freedom)
  sorted-idx    ckpt values, every row's index stream sorted along the columns (clustered, low-entropy indices)
  low-entropy   ckpt values, indices drawn from 16 of the 256 entries (perm-clustered codebook use)
  big-bias      bias N(0, 0.05): column means as large as 2.5 column stds (an un-centred weight matrix)
  ref-test      the reference test's normal(0.02, 0.5) for every tensor (bias as large as the scaled weights); llm-r4: the
                "llm" distribution of rounds 1 - 4 (centroids N(0, 0.02), residual N(0, 0.005), scale 1 + 0.1 N, bias 0.01 N)
Activations rotate per layer: N(0, 1); N(0, 1) with 0.5 % of the channels x 50 (massive activations); |N(0, 1)| (post-ReLU
like, large mean); 90 % zeros; N(0, 1) projected orthogonal to the bias.

Reference on the GPU as in tools/gpu_fuzz_count.py: `vptq_dequant` (bit-identical to the reference CPU path's dequant:
sha256 of W on every golden in tests/test_hip_parity.py) and a float64 product rounded once.  Per family: layers, how
many the gate sent to the reference's roundings, worst / p99.9 / median of max|dy| / max|y_ref| through the product
route, count above the bar (1e-3 fp16, 8e-3 bf16); for information the folded form WITHOUT the gate and the un-rounded
distance the gate measured on its probes (median / max)."""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vptq_amd  # noqa: E402
from vptq_amd import _backend as B  # noqa: E402

SHAPES = [(4096, 4096), (4096, 4096), (4096, 1024), (8192, 8192), (4096, 14336), (14336, 4096), (5120, 5120), (2048, 8192),
          (8192, 1024), (1024, 4096)]   # (in, out)
FAMILIES = ("ckpt", "heavy-scale", "outlier-cols", "zero-bias", "t-centroids", "sorted-idx", "low-entropy", "big-bias", "ref-test", "llm-r4")
XKINDS = ("normal", "massive", "relu", "sparse", "bias-orth")


def make(I, O, fam, dt, dev, g, k=256, kr=256):
    m = vptq_amd.VQuantLinear(I, O, vector_lens=[-1, 8], num_centroids=[-1, k], num_res_centroids=[-1, kr],
                              group_num=1, group_size=I, outlier_size=0, indices_as_float=False, enable_norm=True,
                              enable_perm=False, is_indice_packed=True, bias=False, dtype=dt, device=dev,
                              enable_proxy_error=False)
    rn = lambda *s: torch.randn(*s, generator=g, device=dev)  # noqa: E731
    cw, rw = m.centroids.weight.shape, m.res_centroids.weight.shape
    if fam == "t-centroids":
        t3 = lambda *s: rn(*s) / torch.sqrt((rn(3, *s) ** 2).mean(0))  # noqa: E731
        c, r = t3(*cw), 0.25 * t3(*rw)
    else:
        c, r = rn(*cw), 0.25 * rn(*rw)
    sig = {"heavy-scale": 1.0}.get(fam, 0.3)
    scale = 0.02 * torch.exp(sig * rn(I))
    if fam == "outlier-cols":
        hot = torch.randperm(I, generator=g, device=dev)[: max(1, I // 200)]
        scale[hot] *= 10 + 40 * torch.rand(hot.numel(), generator=g, device=dev)
    bias = 0.002 * rn(I)
    if fam == "zero-bias":
        bias = bias * (0.0 if int(torch.randint(0, 2, (1,), generator=g, device=dev)) else 5e-3)
    if fam == "big-bias":
        bias = 0.05 * rn(I)
    if fam == "ref-test":
        c, r, scale, bias = 0.02 + 0.5 * rn(*cw), 0.02 + 0.5 * rn(*rw), 0.02 + 0.5 * rn(I), 0.02 + 0.5 * rn(I)
    if fam == "llm-r4":
        c, r, scale, bias = 0.02 * rn(*cw), 0.005 * rn(*rw), 1 + 0.1 * rn(I), 0.01 * rn(I)
    m.centroids.weight.data = c.to(dt)
    m.res_centroids.weight.data = r.to(dt)
    m.weight_scale.data = scale.to(dt)
    m.weight_bias.data = bias.to(dt)
    N = m.indices.shape[1]
    if fam in ("sorted-idx", "low-entropy"):
        hi = 16 if fam == "low-entropy" else k
        e = torch.randint(0, hi, (N, I), generator=g, device=dev, dtype=torch.int64)
        er = torch.randint(0, kr, (N, I), generator=g, device=dev, dtype=torch.int64)
        if fam == "sorted-idx":
            e = torch.sort(e, dim=1).values
        w16 = e | (er << 8)
        m.indices.data = (w16[:, 0::2] | (w16[:, 1::2] << 16)).to(torch.int32).view(m.indices.shape)
    else:
        m.indices.data = torch.randint(-2**31, 2**31 - 1, m.indices.shape, generator=g, device=dev, dtype=torch.int64).to(torch.int32)
    return m.eval()


def make_x(m, kind, dt, dev, g):
    I = m.in_features
    x = torch.randn(I, generator=g, device=dev)
    if kind == "massive":
        hot = torch.randperm(I, generator=g, device=dev)[: max(1, I // 200)]
        x[hot] *= 50
    elif kind == "relu":
        x = x.abs()
    elif kind == "sparse":
        x = x * (torch.rand(I, generator=g, device=dev) < 0.1)
    elif kind == "bias-orth":
        b = m.weight_bias.data.float()
        bb = (b * b).sum()
        if float(bb) > 0:
            x = x - (x * b).sum() / bb * b
    return x.to(dt).view(1, 1, I)


def main_chain(a):
    """the same families and activation kinds through the persistent chain launch, one arithmetic per run"""
    from vptq_amd.ops.chain import GemvChain
    dev = torch.device("cuda", 0)
    dt, bar = (torch.float16, 1e-3) if a.dtype == "f16" else (torch.bfloat16, 8e-3)
    # (descriptor flags: the call's flags decide on top of the load-time gate's EXACT - the gate of the arithmetic under test:
    # 6.5e-4 of probe distance for the selective form, 7.5e-4 for the folded one; chain-exact: no gate needed)
    B.set_arithmetic("selective" if a.route == "chain-selective" else "folded")
    fl = {"chain-selective": B.GEMV_SELECTIVE, "chain-folded": 0, "chain-exact": B.GEMV_EXACT}[a.route] | B.GEMV_FORCE_MFMA
    g = torch.Generator(device=dev).manual_seed(7000 + a.seed)
    rng = np.random.default_rng(a.seed)
    shapes = [s for s in SHAPES if s[0] * s[1] <= a.max_elems]
    per = (a.layers + len(FAMILIES) - 1) // len(FAMILIES)
    t0 = time.time()
    total, done = 0, 0
    rows = []
    for fam in FAMILIES:
        errs, bitid, gated, probes = [], [], 0, []
        i = 0
        while i < per:
            nb = min(8, per - i)
            ms, xs = [], []
            for j in range(nb):
                I, O = shapes[int(rng.integers(0, len(shapes)))]
                m = make(I, O, fam, dt, dev, g)
                ms.append(m)
                xs.append(make_x(m, XKINDS[(done + i + j) % len(XKINDS)], dt, dev, g))
            ch = GemvChain(ms)
            name = ch.kernel_name(flags=fl)
            # ("grouped": every layer of this list was turned to the reference's roundings by its load-time gate and the chain kernel has
            # no bf16 form of them - the list then goes out as grouped launches)
            assert name in ("gemv_k256c_kernel", "grouped"), name
            gated += sum(int(bool(m._descriptor()[9] & B.GEMV_EXACT)) for m in ms)
            if a.probe:
                probes += [float(B.folded_probe_distance(m._descriptor()[1], m.in_features, m.out_features, m.weight_bias.data, dt, dev)) for m in ms]
            ys = ch(xs, flags=fl)
            for m, x, y in zip(ms, xs, ys):
                W = m.dequant()
                r16 = (W.double() @ x.reshape(-1).double()).to(dt)
                den = r16.double().abs().max().clamp_min(1e-30)
                errs.append(float((y.reshape(-1).double() - r16.double()).abs().max() / den))
                bitid.append(float((y.reshape(-1).view(torch.int16) == r16.view(torch.int16)).float().mean()))
                del W
            del ms, xs, ch, ys
            i += nb
        done += per
        e = np.array(errs)
        ex = int((e > bar).sum())
        total += ex
        rows.append((fam, len(e), gated, e.max(), np.quantile(e, 0.999), np.median(e), ex, float(np.median(bitid)), float(np.min(bitid))))
        if a.probe:
            pr = np.array(probes)
            print(f"# {fam}: probe distance quantiles 50/90/99/max = {np.quantile(pr, 0.5):.2e} {np.quantile(pr, 0.9):.2e} {np.quantile(pr, 0.99):.2e} {pr.max():.2e}; "
                  f"share above 5e-4 / 5.5e-4 / 6e-4 / 6.5e-4 / 7e-4: " + " ".join(f"{float((pr > t).mean()):.3f}" for t in (5e-4, 5.5e-4, 6e-4, 6.5e-4, 7e-4))
                  + "; layers above the bar (error, probe distance): " + str([(f"{e[k]:.2e}", f"{pr[k]:.2e}") for k in np.nonzero(e > bar)[0]]), flush=True)
        print(f"# {fam}: {per} layers done after {time.time() - t0:.0f} s", flush=True)
    print(f"route {a.route} (one persistent launch per 8 layers), dtype {a.dtype}, bar {bar:g}, {done} layers, seed {a.seed}")
    print(f"{'family':14s} {'layers':>6s} {'gated':>6s} {'worst':>9s} {'p99.9':>9s} {'median':>9s} {'> bar':>6s} {'bit-identical: median':>22s} {'min':>6s}")
    for r in rows:
        print(f"{r[0]:14s} {r[1]:6d} {r[2]:6d} {r[3]:9.2e} {r[4]:9.2e} {r[5]:9.2e} {r[6]:6d} {r[7]:22.4f} {r[8]:6.4f}")
    print(f"exceedances: {total}")
    return 1 if total else 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=4096)
    ap.add_argument("--dtype", default="f16", choices=["f16", "bf16"])
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--max-elems", type=float, default=70e6, help="largest in x out drawn (time)")
    ap.add_argument("--route", default="module", choices=["module", "chain-selective", "chain-folded", "chain-exact"],
                    help="module: VQuantLinear.forward in the process's arithmetic; chain-*: batches of 8 layers through ONE persistent "
                         "launch (vptq_quant_gemv_chain, FORCE_MFMA) with VPTQ_GEMV_SELECTIVE / no flag / VPTQ_GEMV_EXACT (round 6)")
    ap.add_argument("--probe", action="store_true", help="chain routes: also the load-time gate's probe distance of every layer")
    a = ap.parse_args()
    if a.route != "module":
        return main_chain(a)
    dev = torch.device("cuda", 0)
    dt = torch.float16 if a.dtype == "f16" else torch.bfloat16
    bar = 1e-3 if a.dtype == "f16" else 8e-3
    g = torch.Generator(device=dev).manual_seed(7000 + a.seed)
    rng = np.random.default_rng(a.seed)
    shapes = [s for s in SHAPES if s[0] * s[1] <= a.max_elems]
    stats = {}
    t0 = time.time()
    per = (a.layers + len(FAMILIES) - 1) // len(FAMILIES)
    done = 0
    for fam in FAMILIES:
        st = stats.setdefault(fam, dict(err=[], raw=[], gated=0, probe=[]))
        for i in range(per):
            I, O = shapes[int(rng.integers(0, len(shapes)))]
            m = make(I, O, fam, dt, dev, g)
            x = make_x(m, XKINDS[(done + i) % len(XKINDS)], dt, dev, g)
            d = m._descriptor()
            gated = bool(d[9] & B.GEMV_EXACT) if vptq_amd.arithmetic() != "reference" else False   # (sent to the reference roundings by the load-time gate)
            st["gated"] += int(gated)
            W = m.dequant()
            s64 = W.double() @ x.reshape(-1).double()
            r16 = s64.to(dt)
            den = r16.double().abs().max().clamp_min(1e-30)
            y = m(x)                                                  # the product route (with the gate)
            st["err"].append(float((y.reshape(-1).double() - r16.double()).abs().max() / den))
            yr = torch.empty(1, 1, O, dtype=dt, device=dev)          # the folded form whatever the gate says (information)
            B.check(B.lib().vptq_quant_gemv(d[1], x.data_ptr(), yr.data_ptr(), 1, 0, None, 0, B.current_stream_ptr(dev)), "gemv")
            st["raw"].append(float((yr.reshape(-1).double() - r16.double()).abs().max() / den))
            st["probe"].append(float(B.folded_probe_distance(d[1], I, O, m.weight_bias.data, dt, dev)))
            del m, W
        done += per
        print(f"# {fam}: {per} layers done after {time.time() - t0:.0f} s", flush=True)
    print(f"dtype {a.dtype}, bar {bar:g}, {done} layers, seed {a.seed}, arithmetic of the product route: {vptq_amd.arithmetic()}"
          f" (gate line of the folded form: {B.FOLDED_MAX_PROBE_DISTANCE[dt]:g}, un-rounded distance on the probes)")
    print(f"{'family':14s} {'layers':>6s} {'gated':>6s} | product route: {'worst':>9s} {'p99.9':>9s} {'median':>9s} {'> bar':>6s} | "
          f"folded, no gate: {'worst':>9s} {'> bar':>6s} | probe distance: {'median':>9s} {'max':>9s}")
    total = 0
    for fam, st in stats.items():
        e, r, p = np.array(st["err"]), np.array(st["raw"]), np.array(st["probe"])
        ex = int((e > bar).sum())
        total += ex
        print(f"{fam:14s} {len(e):6d} {st['gated']:6d} | {'':14s} {e.max():9.2e} {np.quantile(e, 0.999):9.2e} {np.median(e):9.2e} {ex:6d} | "
              f"{'':16s} {r.max():9.2e} {int((r > bar).sum()):6d} | {'':15s} {np.median(p):9.2e} {p.max():9.2e}")
    print(f"exceedances through the product route: {total}")
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main())
