#!/usr/bin/env python3
"""Randomised parity run of the canonical-format GEMV kernels against the numpy oracle (GPU box).

    python tools/gpu_fuzz.py [--cases 40] [--seed 0]

Random widths (multiples of 8 up to 30000), heights, permutation / output bias, every kernel
and arithmetic flag combination; prints the worst max-normalised error per flag set.
--formats: random index formats (vector length, codebook sizes, groups, outlier columns, ...) instead.
--adversarial: families built against the folded arithmetic (bias-dominated layers, large-mean or bias-orthogonal
activations, the reference test's cyclic indices): worst error per family of the module's default route (with its
load-time gate), the chain route, the folded form itself and the reference's roundings.
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from oracle import vptq_oracle as vo  # noqa: E402  (checker only)
from _cases import rel_err  # noqa: E402
from _gpu_util import spec_to_module, bits_to_tensor, tensor_to_bits, gemv_abi, kernel_name  # noqa: E402

EXACT = 4
FLAGS = {"default": 0, "exact": 4, "mfma": 8, "mfma+exact": 12, "valu": 16, "valu+exact": 20}


def fuzz_formats(a, dev):
    """Random index FORMATS (vector length, codebook sizes, codebook groups, outlier columns, norm, perm, bias,
    1-6 tokens) through the library's kernel choice and through the generic kernel, against the oracle."""
    rng = np.random.default_rng(a.seed)
    used, worst = {}, 0.0
    for c in range(a.cases):
        v = int(rng.choice([8, 8, 12, 16, 16, 4, 6, 2, 10]))
        ib = int(rng.integers(4, 17))
        rb = int(rng.choice([0, 0, int(rng.integers(2, 17))]))
        if ib + rb > 32:
            rb = 32 - ib
        C = int(rng.choice([1, 1, 1, 2, 3]))
        G = 4 * int(rng.integers(1, 300)) if rng.integers(0, 4) else 2 * int(rng.integers(1, 300))
        S = int(rng.choice([0, 0, 4 * int(rng.integers(1, 40))]))
        O = int(rng.integers(1, 40)) * v - int(rng.integers(0, v))
        O = max(O, 1)
        kw = dict(vector_len=v, num_centroids=1 << ib, num_res_centroids=(1 << rb) if rb else 0, num_codebooks=C,
                  enable_perm=bool(rng.integers(0, 2)), enable_norm=bool(rng.integers(0, 4)), bias=bool(rng.integers(0, 2)))
        if S:
            kw.update(outlier_size=S, outlier_vector_len=v if rng.integers(0, 3) else int(rng.choice([4, 8])),
                      num_outlier_centroids=1 << int(rng.integers(2, 11)))
        I = S + C * G
        dt = a.dtype
        tol = 1e-3 if dt == "f16" else 8e-3
        tokens = int(rng.integers(1, 7))
        try:
            L = vo.make_layer(I, O, dist="llm" if rng.integers(0, 2) else "ref-test", seed=2000 + c, dtype=dt, **kw)
        except AssertionError:
            continue
        x = vo.from_f32((0.3 * rng.standard_normal((1, tokens, I))).astype(np.float32), dt)
        m = spec_to_module(L, dev)
        xt = bits_to_tensor(x, dt, dev).reshape(x.shape)
        want = vo.gemv(vo.dequant(L, ref_residual_mask_quirk=False), x, dt, L.bias)
        kn = kernel_name(m, tokens)
        used[kn] = used.get(kn, 0) + 1
        line = f"case {c:3d} v={v} k=2^{ib} kr=2^{rb} C={C} G={G} S={S} O={O} tok={tokens} {kn}:"
        for name, fl in (("default", 0), ("generic", 2)):
            e = rel_err(tensor_to_bits(gemv_abi(m, xt, fl)), want, dt)
            worst = max(worst, e)
            line += f" {name}={e:.1e}"
            assert e <= tol, (line, kw)
        print(line, flush=True)
    print("kernels used:", used, "worst:", f"{worst:.2e}")


def fuzz_lds_tall(a, dev):
    """Tall layers (>= 1024 vector-rows) of the LDS-resident formats, one token: gemv_lds_mfma_kernel (folded form)
    against the C oracle, the kernel with the reference's roundings and the generic kernel."""
    from oracle import c_oracle as co
    rng = np.random.default_rng(a.seed)
    used, worst = {}, 0.0
    fmts = [(4096, 0), (8192, 0), (4096, 256), (8192, 256), (4096, 512), (2048, 512), (8192, 512), (1024, 4), (512, 16)]
    for c in range(a.cases):
        k, kr = fmts[int(rng.integers(0, len(fmts)))]
        I = 8 * int(rng.integers(2, 1100))
        O = 8192 + int(rng.choice([0, 8, 24, 8 * int(rng.integers(1, 1100)), int(rng.integers(1, 64))]))
        kw = dict(num_centroids=k, num_res_centroids=kr, enable_perm=bool(rng.integers(0, 2)),
                  enable_norm=bool(rng.integers(0, 4)), bias=bool(rng.integers(0, 2)))
        dt = a.dtype
        tol = 1e-3 if dt == "f16" else 8e-3
        L = vo.make_layer(I, O, dist="llm" if rng.integers(0, 2) else "ref-test", seed=3000 + c, dtype=dt, **kw)
        x = vo.from_f32((0.3 * rng.standard_normal((1, 1, I))).astype(np.float32), dt)
        m = spec_to_module(L, dev)
        xt = bits_to_tensor(x, dt, dev).reshape(x.shape)
        want = co.forward(L, x, quirk=False)
        kn = kernel_name(m, 1)
        used[kn] = used.get(kn, 0) + 1
        line = f"case {c:3d} k={k} kr={kr} I={I} O={O} perm={int(kw['enable_perm'])} norm={int(kw['enable_norm'])} bias={int(kw['bias'])} {kn}:"
        for name, fl in (("default", 0), ("exact", 4), ("generic", 2)):
            e = rel_err(tensor_to_bits(gemv_abi(m, xt, fl)), want, dt)
            worst = max(worst, e)
            line += f" {name}={e:.1e}"
            assert e <= tol, (line, kw)
        print(line, flush=True)
    print("kernels used:", used, "worst:", f"{worst:.2e}")


def fuzz_adversarial(a, dev):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_hip_parity as T
    from vptq_amd.ops.chain import GemvChain
    rng = np.random.default_rng(a.seed)
    worst = {}
    fams = T.ADVERSARIAL + [("bias2", "orthogonal"), ("bias3", "orthogonal"), ("bias8", "normal"), ("bias64", "orthogonal")]
    for c in range(a.cases):
        family, xkind = fams[c % len(fams)]
        I = 8 * int(rng.integers(64, 1100))
        O = 8 * int(rng.choice([int(rng.integers(8, 100)), int(rng.integers(576, 700))]))
        L = T._adversarial_layer(I, O, family, seed=5000 + c)
        x = T._adversarial_x(L, xkind, seed=9000 + c)
        want = vo.forward(L, x)
        m = spec_to_module(L, dev)
        xt = bits_to_tensor(x, "f16", dev).reshape(x.shape)
        gated = m._descriptor()[9] != 0
        errs = {"default_route": rel_err(tensor_to_bits(m(xt)), want, "f16"),
                "chain_route": rel_err(tensor_to_bits(GemvChain([m, m])([xt, xt], flags=None if gated else 8)[1]), want, "f16"),
                "folded_form": rel_err(tensor_to_bits(gemv_abi(m, xt, 0)), want, "f16"),
                "reference_roundings": rel_err(tensor_to_bits(gemv_abi(m, xt, 4)), want, "f16")}
        key = f"{family}/{xkind}"
        w = worst.setdefault(key, {k: 0.0 for k in errs} | {"gated": gated})
        for k, v in errs.items():
            w[k] = max(w[k], v)
        print(f"case {c:3d} I={I:5d} O={O:5d} {key:22s} gated={int(gated)} " + " ".join(f"{k}={v:.1e}" for k, v in errs.items()), flush=True)
        assert errs["default_route"] <= 1e-3 and errs["chain_route"] <= 1e-3, key
    print("worst per family:")
    for k, w in worst.items():
        print(f"  {k:22s} gated={int(w['gated'])} " + " ".join(f"{n}={v:.2e}" for n, v in w.items() if n != "gated"))


B_EXACT, B_MFMA = 1 << 2, 1 << 3   # VPTQ_GEMV_EXACT, VPTQ_GEMV_FORCE_MFMA


def fuzz_tokens(a, dev):
    """random canonical layers x 2 ... 40 tokens through the library's default route with the scratch buffer
    (gemv_k256m for 2-4 tokens, gemm_k256t from 5: partial sweeps / row groups, perm, bias, launches of 16) against the
    oracle; the same call twice - bit-identical"""
    rng = np.random.default_rng(a.seed)
    dt = a.dtype
    tol = 1e-3 if dt == "f16" else 8e-3
    worst = {}
    for c in range(a.cases):
        I = int(rng.choice([8 * int(rng.integers(16, 1800)), 2048 * int(rng.integers(1, 8)), 2048 * int(rng.integers(1, 8)) + 8]))
        O = int(rng.choice([8 * int(rng.integers(1, 200)), 8 * int(rng.integers(200, 1400)) - int(rng.integers(0, 8))]))
        O = max(O, 8)
        if I * O > 16e6:
            O = max(8, int(16e6 // I) // 8 * 8)
        tokens = int(rng.choice([int(rng.integers(2, 5)), int(rng.integers(5, 17)), int(rng.integers(17, 41))]))
        kw = dict(enable_perm=bool(rng.integers(0, 2)), bias=bool(rng.integers(0, 2)))
        L = vo.make_layer(I, O, dist="llm", seed=3000 + c, dtype=dt, **kw)
        x = vo.from_f32(rng.standard_normal((1, tokens, I)).astype(np.float32), dt)
        m = spec_to_module(L, dev)
        xt = bits_to_tensor(x, dt, dev).reshape(x.shape)
        want = vo.forward(L, x)
        # --exact: the reference's roundings (the product's default arithmetic): gemv_k256m / gemv_k256 for 2-4 tokens, gemm_k256 from 5;
        # O is drawn tall now and then so that the persistent kernel's 144-row-group threshold is crossed
        fl = (B_EXACT | (B_MFMA if (a.exact and c % 3 == 0 and tokens <= 4) else 0)) if a.exact else 0
        name = kernel_name(m, min(tokens, 16), fl)
        got = gemv_abi(m, xt, fl)
        torch.cuda.synchronize()
        got2 = gemv_abi(m, xt, fl)
        torch.cuda.synchronize()
        assert torch.equal(got.view(torch.int16), got2.view(torch.int16)), (c, "not reproducible")
        e = rel_err(tensor_to_bits(got), want, dt)
        worst[name] = max(worst.get(name, 0.0), e)
        print(f"case {c:3d} I={I:6d} O={O:6d} tokens={tokens:2d} perm={int(kw['enable_perm'])} bias={int(kw['bias'])} {name}: {e:.2e}", flush=True)
        assert e <= tol, (c, I, O, tokens, name, e)
    print("worst:", {k: f"{v:.2e}" for k, v in worst.items()})


def fuzz_chains(a, dev):
    """random chains through the persistent launch (vptq_quant_gemv_chain / gemv_k256c_kernel): 2-14 layers of random
    shapes (partial sweeps / row groups, more or fewer row groups than workgroups, bias), independent and dependent,
    every layer against the one-launch-per-layer EXACT kernel (<= tol) and - same chain called twice - bit-identical"""
    from vptq_amd.ops.chain import GemvChain
    from vptq_amd import _backend as B
    rng = np.random.default_rng(a.seed)
    dt = a.dtype
    tol = 1e-3 if dt == "f16" else 8e-3
    CHAIN = B.GEMV_FORCE_MFMA   # the chain kernel also where the layers do not fill the device
    worst = 0.0
    for c in range(a.cases):
        dependent = bool(rng.integers(0, 3) == 0)
        n = int(rng.integers(2, 15))
        if dependent:
            dims = [8 * int(rng.integers(16, 700)) for _ in range(n + 1)]
            shapes = [(dims[i], dims[i + 1]) for i in range(n)]
        else:
            shapes = []
            for _ in range(n):
                I = int(rng.choice([8 * int(rng.integers(16, 1200)), 2048 * int(rng.integers(1, 5)), 2048 * int(rng.integers(1, 5)) + 8]))
                O = int(rng.choice([8 * int(rng.integers(1, 120)), 8 * int(rng.integers(120, 1100))]))
                shapes.append((I, O))
        Ls = [vo.make_layer(I, O, dist="llm", seed=7000 + 31 * c + i, dtype=dt, bias=bool(rng.integers(0, 2))) for i, (I, O) in enumerate(shapes)]
        ms = [spec_to_module(L, dev) for L in Ls]
        xs = [bits_to_tensor(vo.from_f32(rng.standard_normal((1, 1, I)).astype(np.float32), dt), dt, dev).reshape(1, 1, I)
              for (I, _) in shapes]
        chain = GemvChain(ms, dependent=dependent)
        name = chain.kernel_name(1, CHAIN)
        ys = chain([xs[0]] if dependent else xs, flags=CHAIN)
        torch.cuda.synchronize()
        ys2 = chain([xs[0]] if dependent else xs, flags=CHAIN)
        torch.cuda.synchronize()
        errs = []
        xin = xs[0]
        for i, (m, y, y2) in enumerate(zip(ms, ys, ys2)):
            assert torch.equal(y.view(torch.int16), y2.view(torch.int16)), (c, i, "not reproducible")
            ref = gemv_abi(m, xin if dependent else xs[i], EXACT)
            errs.append(rel_err(tensor_to_bits(y), tensor_to_bits(ref), dt))
            xin = y
        worst = max(worst, max(errs))
        print(f"case {c:3d} {'dep' if dependent else 'ind'} n={n:2d} {name}: max err {max(errs):.2e}  shapes {shapes[:3]}...", flush=True)
        assert max(errs) <= tol, (c, errs, shapes)
    print(f"worst: {worst:.2e}")


def fuzz_sliced(a, dev):
    """random layers of the k >= 16384 family (v = 8 / 16; 16384 ... 65536 main centroids; residual codebook none, 256 or any
    power of two up to 65536), one token, over the sliced layout(s) (gemv_sliced_kernel - the module's default one-token route
    for these formats): against the oracle and the gather kernels, twice (reproducible bits), with random rows-per-wave;
    skewed index distributions (most elements in one slice, empty slices - of the main AND the residual table) included"""
    from vptq_amd.utils.sliced import SlicedGemv
    from vptq_amd.utils.pack import pack_index
    from vptq_amd import _backend as B_
    rng = np.random.default_rng(a.seed)
    dt = a.dtype
    tol = 1e-3 if dt == "f16" else 8e-3
    worst = 0.0
    for c in range(a.cases):
        v = int(rng.choice([8, 8, 16]))
        k = int(rng.choice([65536, 65536, 65536, 32768, 16384]))
        kr = int(rng.choice([0, 256, 65536, 4, 64, 1024, 4096, 16384, 32768, 2]))
        I = int(rng.choice([8 * int(rng.integers(8, 1800)), 2048 * int(rng.integers(1, 8)), 8 * int(rng.integers(1800, 3600)), 48 * int(rng.integers(340, 600))]))
        O = int(rng.choice([v * int(rng.integers(33, 80)), v * int(rng.integers(64, 700)) - int(rng.integers(0, v))]))
        L = vo.make_layer(I, O, dist="llm", seed=9000 + c, dtype=dt, vector_len=v, num_centroids=k, num_res_centroids=kr,
                          bias=bool(rng.integers(0, 2)), enable_perm=bool(rng.integers(0, 3) == 0))
        skew = int(rng.integers(0, 4))
        if skew:   # rewrite the indices: 1 = one slice only, 2 = two slices, 3 = 90 % in one slice (main and residual alike)
            N = L.indices.shape[1]
            ib, rb = L.index_bits, L.res_bits

            def skewed(bits):
                if bits == 0:
                    return None
                idx = rng.integers(0, 1 << bits, size=(1, N, I), dtype=np.int64)
                top = max(bits - 3, 0)
                low = (1 << top) - 1
                if skew == 1:
                    idx = (idx & low) | ((3 % (1 << (bits - top))) << top)
                elif skew == 2:
                    idx = (idx & low) | ((rng.integers(0, 2, size=idx.shape) * (7 % (1 << (bits - top)))) << top)
                else:
                    idx = np.where(rng.random(idx.shape) < 0.9, (idx & low) | ((5 % (1 << (bits - top))) << top), idx)
                return idx
            L.indices = vo.pack_indices(skewed(ib), ib, skewed(rb), rb)
        x = vo.from_f32(rng.standard_normal((1, 1, I)).astype(np.float32), dt)
        print(f"case {c:3d} start: v{v}-k{k}-{kr} I={I} O={O} skew={skew} perm={L.perm is not None} bias={L.bias is not None}", flush=True)
        m = spec_to_module(L, dev)
        m.enable_sliced_layout(False)
        xt = bits_to_tensor(x, dt, dev).reshape(x.shape)
        rpw = int(rng.choice([0, 1, 2, 3, 5]))
        sl = SlicedGemv(m, rows_per_wave=rpw)
        got = sl(xt)
        torch.cuda.synchronize()
        print(f"  folded ok (rpw {rpw}, slices {sl.slices})", flush=True)
        again = sl(xt)
        assert torch.equal(got.view(torch.int16), again.view(torch.int16)), (c, "not reproducible")
        # (residual indices masked with res_bits as the reference's CUDA kernel does, quant_gemv.cuh:120-121: the Python
        # path's mask differs where res_bits > index_bits, e.g. k16384 + 32768 - DESIGN.md section 3)
        want = vo.gemv(vo.dequant(L, ref_residual_mask_quirk=False), x, dt, L.bias)
        e = rel_err(tensor_to_bits(got), want, dt)
        e2 = rel_err(tensor_to_bits(got), tensor_to_bits(gemv_abi(m, xt, 0)), dt)
        worst = max(worst, e)
        # the reference's roundings over a layout of their own (round 5: ONE layout; any residual codebook but v8's 256-entry
        # one gathered from device memory), where the library serves the layer: almost every output bit-identical
        ex = -1.0
        sx = None
        from vptq_amd.utils.sliced import exact_column_parts
        if exact_column_parts(m._descriptor()[1], I)[0]:    # (in one piece, or - layers wider than ~16300 columns - as 2 / 3 column parts)
            sx = SlicedGemv(m, rows_per_wave=rpw, exact=True)
            print(f"  exact layout built (slices {sx.slices}, column parts {sx.parts})", flush=True)
            gx = sx(xt)
            torch.cuda.synchronize()
            assert torch.equal(gx.view(torch.int16), sx(xt).view(torch.int16)), (c, "exact: not reproducible")
            ex = rel_err(tensor_to_bits(gx), want, dt)
            ident = float((tensor_to_bits(gx).reshape(-1) == np.asarray(want).reshape(-1)).mean())
            assert ex <= tol and ident >= 0.9, (c, v, k, kr, I, O, ex, ident)
            worst = max(worst, ex)
        # 2 - 4 tokens in one launch over the same layouts (gemv_sliced_tok.hip: column phases), where the library takes it
        T = int(rng.integers(2, 5))
        et = -1.0
        if sl.tokens_supported(T):
            xT = vo.from_f32(rng.standard_normal((1, T, I)).astype(np.float32), dt)
            xTt = bits_to_tensor(xT, dt, dev).reshape(xT.shape)
            gotT = sl.forward_tokens(xTt)
            torch.cuda.synchronize()
            assert torch.equal(gotT.view(torch.int16), sl.forward_tokens(xTt).view(torch.int16)), (c, "tokens: not reproducible")
            et = rel_err(tensor_to_bits(gotT), vo.gemv(vo.dequant(L, ref_residual_mask_quirk=False), xT, dt, L.bias), dt)
            worst = max(worst, et)
        # ... and 2 - 8 tokens in the reference's roundings over the exact layout (gemv_sliced_tok.hip, EX: one-table formats)
        TX = int(rng.integers(2, 9))
        etx = -1.0
        if sx is not None and sx.tokens_supported(TX):
            xT = vo.from_f32(rng.standard_normal((1, TX, I)).astype(np.float32), dt)
            xTt = bits_to_tensor(xT, dt, dev).reshape(xT.shape)
            gX = sx.forward_tokens(xTt)
            torch.cuda.synchronize()
            assert torch.equal(gX.view(torch.int16), sx.forward_tokens(xTt).view(torch.int16)), (c, "exact tokens: not reproducible")
            wX = vo.gemv(vo.dequant(L, ref_residual_mask_quirk=False), xT, dt, L.bias)
            etx = rel_err(tensor_to_bits(gX), wX, dt)
            identx = float((tensor_to_bits(gX).reshape(-1) == np.asarray(wX).reshape(-1)).mean())
            assert etx <= tol and identx >= 0.9, (c, v, k, kr, I, O, TX, etx, identx)
            worst = max(worst, etx)
        print(f"case {c:3d} v{v}-k{k}-{kr} I={I:6d} O={O:5d} skew={skew} slices={sl.slices} tables={len(sl.layout)} "
              f"whole={sl._whole} rpw={sl.layout[0].rows_per_wave}: oracle {e:.2e} gather {e2:.2e} | {T} tokens {et:.2e} | reference roundings {ex:.2e}"
              f" | {TX} tokens in them {etx:.2e}", flush=True)
        assert e <= tol and e2 <= tol and et <= tol, (c, e, e2, et)
    print(f"worst: {worst:.2e}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=40)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--dtype", default="f16", choices=["f16", "bf16"])
    ap.add_argument("--formats", action="store_true", help="random index formats instead of the canonical one")
    ap.add_argument("--lds-tall", action="store_true", help="tall layers of the LDS-resident formats (gemv_lds_mfma_kernel)")
    ap.add_argument("--adversarial", action="store_true", help="families built against the folded arithmetic")
    ap.add_argument("--tokens", action="store_true", help="random token counts 2 ... 40 through the default route")
    ap.add_argument("--exact", action="store_true", help="--tokens: with VPTQ_GEMV_EXACT (every third small-token case also with FORCE_MFMA)")
    ap.add_argument("--chains", action="store_true", help="random chains through the persistent chain launch")
    ap.add_argument("--sliced", action="store_true", help="random k = 65536 layers over the sliced layout")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    if a.tokens:
        return fuzz_tokens(a, dev)
    if a.chains:
        return fuzz_chains(a, dev)
    if a.sliced:
        return fuzz_sliced(a, dev)
    if a.adversarial:
        return fuzz_adversarial(a, dev)
    if a.lds_tall:
        return fuzz_lds_tall(a, dev)
    if a.formats:
        return fuzz_formats(a, dev)
    rng = np.random.default_rng(a.seed)
    worst = {k: 0.0 for k in FLAGS}
    for c in range(a.cases):
        I = int(rng.choice([8 * int(rng.integers(16, 3750)), 2048 * int(rng.integers(1, 15)),
                            2048 * int(rng.integers(1, 15)) + 8]))
        O = int(rng.choice([8 * int(rng.integers(1, 200)), 8 * int(rng.integers(200, 1400)) - int(rng.integers(0, 8))]))
        O = max(O, 8)
        if I * O > 40e6:
            O = max(8, int(40e6 // I) // 8 * 8)
        kw = dict(enable_perm=bool(rng.integers(0, 2)), bias=bool(rng.integers(0, 2)))
        dt = a.dtype
        tol = 1e-3 if dt == "f16" else 8e-3
        L = vo.make_layer(I, O, dist="llm", seed=1000 + c, dtype=dt, **kw)
        x = vo.from_f32(rng.standard_normal((1, 1, I)).astype(np.float32), dt)
        m = spec_to_module(L, dev)
        xt = bits_to_tensor(x, dt, dev).reshape(x.shape)
        want = vo.forward(L, x)
        line = f"case {c:3d} I={I:6d} O={O:6d} perm={int(kw['enable_perm'])} bias={int(kw['bias'])}:"
        for name, fl in FLAGS.items():
            got = tensor_to_bits(gemv_abi(m, xt, fl))
            e = rel_err(got, want, dt)
            worst[name] = max(worst[name], e)
            line += f" {name}={e:.1e}"
            assert e <= tol, (line, kernel_name(m, 1, fl))
        print(line, flush=True)
    print("worst:", {k: f"{v:.2e}" for k, v in worst.items()})


if __name__ == "__main__":
    main()
