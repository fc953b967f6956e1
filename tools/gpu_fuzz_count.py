#!/usr/bin/env python3
"""Counting fuzz of the DEFAULT (folded) arithmetic of the canonical-format kernels: how often, and by how much, does a
layer's output differ from the reference CPU path's - N >= 2000 layers per dtype instead of a few dozen (VERDICT r3
"what's weak" 1: "nobody has counted the tail").

    python tools/gpu_fuzz_count.py [--layers 2048] [--dtype f16|bf16] [--seed 0] [--spot 24]

Per (value distribution, kernel): number of layers, worst / p99.9 / median of max|dy| / max|y_ref|, the
count above the bar (1e-3 fp16, 8e-3 bf16) and - from the kernels' float32 outputs (VPTQ_GEMV_OUT_F32) - the largest
distance of the UN-ROUNDED sum from the reference's un-rounded sum, max-normalised like the error itself.  The rounded
error is that distance plus at most one rounding flip of the output it sits on (one 16-bit ulp: up to 2^-10 = 9.77e-4
of max|y| in the top binade for fp16, 2^-7 = 7.8e-3 for bf16).  The reference's own per-weight roundings put its
un-rounded sums 1-6e-4 of max|y| away from exact arithmetic (tools/fold_error_study.py), so only an arithmetic that
repeats those roundings (VPTQ_GEMV_EXACT: distance ~1e-6) stays under the bar for every draw.

The reference here is computed on the GPU so that thousands of layers at BASELINE sizes take seconds: `vptq_dequant`
(bit-identical to the reference CPU path's `dequant`: sha256 of W in tests/test_hip_parity.py on every golden) and a
float64 product, rounded once to the 16-bit type - the oracle's arithmetic model (oracle/vptq_oracle.py: float64
accumulation, one rounding).  Every `--spot`-th layer is also checked against the C oracle itself (checker only).

Distributions: llm (centroids N(0, 0.02), residual N(0, 0.005), scale 1 + 0.1 N, bias 0.01 N, x N(0, 1)), ref-test
(the reference test's normal(0.02, 0.5) for everything), cyclic (ref-test values, indices arange(k) as in
tests/test_quant_gemv.py of the reference), large-mean (llm layer, x N(3, 1))."""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import vptq_amd  # noqa: E402
from vptq_amd import _backend as B  # noqa: E402
from vptq_amd.ops.chain import GemvChain  # noqa: E402

SHAPES = [(4096, 4096), (4096, 4096), (8192, 8192), (4096, 1024), (8192, 1024), (4096, 14336), (14336, 4096),
          (5120, 5120), (2048, 8192), (8192, 3072), (11008, 4096), (4096, 11008)]   # (in, out)
DISTS = ("llm", "ref-test", "cyclic", "large-mean")


def make(I, O, dist, dt, dev, g):
    m = vptq_amd.VQuantLinear(I, O, vector_lens=[-1, 8], num_centroids=[-1, 256], num_res_centroids=[-1, 256],
                              group_num=1, group_size=I, outlier_size=0, indices_as_float=False, enable_norm=True,
                              enable_perm=False, is_indice_packed=True, bias=False, dtype=dt, device=dev,
                              enable_proxy_error=False)
    rn = lambda *s: torch.randn(*s, generator=g, device=dev)  # noqa: E731
    if dist in ("llm", "large-mean"):
        m.centroids.weight.data = (rn(*m.centroids.weight.shape) * 0.02).to(dt)
        m.res_centroids.weight.data = (rn(*m.res_centroids.weight.shape) * 0.005).to(dt)
        m.weight_scale.data = (1 + 0.1 * rn(I)).to(dt)
        m.weight_bias.data = (0.01 * rn(I)).to(dt)
        x = (rn(1, 1, I) + (3.0 if dist == "large-mean" else 0.0)).to(dt)
    else:
        m.centroids.weight.data = (0.02 + 0.5 * rn(*m.centroids.weight.shape)).to(dt)
        m.res_centroids.weight.data = (0.02 + 0.5 * rn(*m.res_centroids.weight.shape)).to(dt)
        m.weight_scale.data = (0.02 + 0.5 * rn(I)).to(dt)
        m.weight_bias.data = (0.02 + 0.5 * rn(I)).to(dt)
        x = (0.02 + 0.5 * rn(1, 1, I)).to(dt)
    if dist == "cyclic":
        # element (n, g) = arange(k) cyclically over the flattened (n, g) order, main = residual index
        e = (torch.arange(m.indices.shape[1] * I, device=dev, dtype=torch.int64) % 256).view(-1, I)
        w16 = e | (e << 8)
        m.indices.data = (w16[:, 0::2] | (w16[:, 1::2] << 16)).to(torch.int32).view(m.indices.shape)
    else:
        m.indices.data = torch.randint(-2**31, 2**31 - 1, m.indices.shape, generator=g, device=dev, dtype=torch.int64).to(torch.int32)
    return m.eval(), x


def reference(m, x):
    """(y rounded to the dtype, un-rounded float64 sums)"""
    W = m.dequant()   # bit-identical to the reference CPU path's dequant
    s = (W.double() @ x.reshape(-1).double())
    return s.to(W.dtype), s


def ulp_of(y16):
    """spacing of the 16-bit type at |y| (float64)"""
    a = y16.double().abs().clamp_min(1e-30)
    e = torch.floor(torch.log2(a))
    mant = 10 if y16.dtype == torch.float16 else 7
    emin = -14 if y16.dtype == torch.float16 else -126
    return torch.pow(2.0, torch.clamp(e, min=emin) - mant)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=2048)
    ap.add_argument("--dtype", default="f16", choices=["f16", "bf16"])
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--spot", type=int, default=24, help="every n-th layer also against the C oracle (0: never)")
    ap.add_argument("--chain", type=int, default=16, help="layers per chain launch")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    dt = torch.float16 if a.dtype == "f16" else torch.bfloat16
    bar = 1e-3 if a.dtype == "f16" else 8e-3
    g = torch.Generator(device=dev).manual_seed(1000 + a.seed)
    rng = np.random.default_rng(a.seed)
    if a.spot:
        from oracle import c_oracle as co
        from _gpu_util import module_to_spec, tensor_to_bits
        from _cases import rel_err
    stats = {}    # (dist, kernel) -> list of (err, dulp)
    names = {}
    spot_worst = 0.0
    t0 = time.time()
    done = 0
    per_dist = (a.layers + len(DISTS) - 1) // len(DISTS)
    for dist in DISTS:
        left = per_dist
        while left > 0:
            I, O = SHAPES[int(rng.integers(0, len(SHAPES)))]
            n = min(a.chain, left)
            layers, xs = zip(*[make(I, O, dist, dt, dev, g) for _ in range(n)])
            refs = [reference(m, x) for m, x in zip(layers, xs)]
            ch = GemvChain(list(layers))
            outs = {}
            kn = ch.kernel_name(1, 0) or "?"
            gated = any(m._descriptor()[9] for m in layers)
            if gated:
                kn += "+gate"
            outs["chain:" + kn] = (ch(list(xs), flags=0), ch(list(xs), flags=B.GEMV_OUT_F32))
            if a.dtype == "f16":   # the reference's roundings inside the chain launch
                kx = ch.kernel_name(1, B.GEMV_EXACT) or "?"
                outs["chain-exact:" + kx] = (ch(list(xs), flags=B.GEMV_EXACT), ch(list(xs), flags=B.GEMV_EXACT | B.GEMV_OUT_F32))
            # single launches with the kernel pinned; like every product route they carry the layer's load-time gate
            # (VQuantLinear._descriptor()[9]); "ungated" = the folded form whatever the gate says (reported, not counted)
            for tag, fl, gate in (("mfma", B.GEMV_FORCE_MFMA, True), ("valu", B.GEMV_FORCE_VALU, True), ("ungated-mfma", B.GEMV_FORCE_MFMA, False)):
                y16, y32 = [], []
                for m, x in zip(layers, xs):
                    d = m._descriptor()
                    fg = fl | (d[9] if gate else 0)
                    name = B.lib().vptq_quant_gemv_kernel_name(d[1], 1, fg)
                    names[tag] = name.decode() if name else "?"
                    for out, f in ((y16, fg), (y32, fg | B.GEMV_OUT_F32)):
                        y = torch.empty(1, 1, O, dtype=torch.float32 if f & B.GEMV_OUT_F32 else dt, device=dev)
                        B.check(B.lib().vptq_quant_gemv(d[1], x.data_ptr(), y.data_ptr(), 1, f, None, 0, B.current_stream_ptr(dev)), "gemv")
                        out.append(y)
                outs[tag + ":" + names[tag]] = (y16, y32)
            # 8 tokens through the one-pass batched-decode kernel (row 0 = the layer's x, the others scaled copies)
            y16 = []
            for m, x in zip(layers, xs):
                x8 = (x.reshape(1, -1).float() * torch.linspace(1.0, 0.3, 8, device=dev).view(8, 1)).to(dt).view(1, 8, I).contiguous()
                d = m._descriptor()
                name = B.lib().vptq_quant_gemv_kernel_name(d[1], 8, 0)
                names["tok8"] = name.decode() if name else "?"
                y16.append(m(x8)[:, :1, :])
            outs["tok8:" + names["tok8"]] = (y16, None)
            torch.cuda.synchronize()
            for key, (l16, l32) in outs.items():
                st = stats.setdefault((dist, key), [])
                for i, (r16, r64) in enumerate(refs):
                    den = r16.double().abs().max().clamp_min(1e-30)
                    err = float(((l16[i].reshape(-1).double() - r16.double()).abs().max() / den))
                    # un-rounded sums (float32 outputs) against the reference's un-rounded sums, max-normalised
                    dulp = float((l32[i].reshape(-1).double() - r64).abs().max() / den) if l32 is not None else float("nan")
                    st.append((err, dulp))
            if a.spot:
                for i, (m, x) in enumerate(zip(layers, xs)):
                    if (done + i) % a.spot == 0 and I * O <= 8192 * 8192:
                        want = co.forward(module_to_spec(m), tensor_to_bits(x).reshape(1, 1, I), quirk=False)
                        e = rel_err(tensor_to_bits(refs[i][0]), np.asarray(want).reshape(-1), a.dtype)
                        spot_worst = max(spot_worst, e)
            done += n
            left -= n
            del layers, xs, refs, ch, outs
        print(f"# {dist}: {per_dist} layers done after {time.time() - t0:.0f} s", flush=True)
    print(f"dtype {a.dtype}, bar {bar:g}, {done} layers, seed {a.seed}; GPU reference vs the C oracle on every {a.spot}-th layer: worst {spot_worst:.2e}")
    print(f"{'distribution':12s} {'route:kernel':60s} {'layers':>6s} {'worst':>9s} {'p99.9':>9s} {'median':>9s} {'> bar':>6s} {'un-rounded: max|d|/max|y|':>26s}")
    total_exceed = 0
    for (dist, key), st in stats.items():
        e = np.array([s[0] for s in st])
        u = np.array([s[1] for s in st])
        ex = int((e > bar).sum())
        if not key.startswith("ungated"):
            total_exceed += ex
        print(f"{dist:12s} {key:60s} {len(e):6d} {e.max():9.2e} {np.quantile(e, 0.999):9.2e} {np.median(e):9.2e} {ex:6d} {np.nanmax(u) if np.isfinite(u).any() else float('nan'):26.2e}")
    print(f"exceedances in all (product routes, i.e. with the load-time gate; the ungated rows are information): {total_exceed}")
    return 1 if total_exceed else 0


if __name__ == "__main__":
    sys.exit(main())
