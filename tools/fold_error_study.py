"""CPU only: where the default (folded) arithmetic's distance to the reference comes from.  numpy emulation of
  folded:      y = sum (c + r) * r16(s x) + sum b x          (what the GPU kernels compute, fp32 sums)
  pre-added:   y = sum r16(c + r) * r16(s x) + sum b x       (the reference's first rounding kept; -DVPTQ_K256C_PREADD)
against the oracle (reference: r16(r16(r16(c + r) s) + b), then x W^T), max-normalised as in the parity tests.
  python tools/fold_error_study.py"""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')):
    sys.path.insert(0, p)
from oracle import vptq_oracle as vo
from _cases import rel_err

def pieces(L):
    dt = L.dtype
    C, G, v, N = L.num_codebooks, L.group_size, L.vector_len, L.num_indices
    idx, ridx = vo.unpack_indices(L.indices, L.index_bits, G, L.res_bits, True)
    cent = vo.to_f32(L.centroids, dt).reshape(C, L.num_centroids, v)
    sel = np.take_along_axis(cent[:, :, None, :], idx.reshape(C, N * G, 1, 1), axis=1).reshape(C, N, G, v)
    q = sel.transpose(1, 3, 0, 2).reshape(N * v, C * G)
    rc = vo.to_f32(L.res_centroids, dt).reshape(C, L.num_res_centroids, v)
    rs = np.take_along_axis(rc[:, :, None, :], ridx.reshape(C, N * G, 1, 1), axis=1).reshape(C, N, G, v).transpose(1, 3, 0, 2).reshape(N * v, C * G)
    return q, rs

for dt, tol in (("f16", 1e-3), ("bf16", 8e-3)):
    ea, eb = [], []
    for seed in range(24):
        rng = np.random.default_rng(seed)
        I = int(rng.choice([2048, 4096, 8192])); O = int(rng.choice([256, 512, 1024]))
        L = vo.make_layer(I, O, dist="llm", seed=500 + seed, dtype=dt, bias=bool(seed & 1))
        x = vo.from_f32(rng.standard_normal((1, 1, I)).astype(np.float32), dt)
        want = vo.forward(L, x)
        q, rs = pieces(L)
        s = vo.to_f32(L.weight_scale, dt).astype(np.float32); b = vo.to_f32(L.weight_bias, dt).astype(np.float32)
        xf = vo.to_f32(x, dt).reshape(-1).astype(np.float32)
        sx = vo.round_to(s * xf, dt).astype(np.float64)
        bx = float((b.astype(np.float64) * xf.astype(np.float64)).sum())
        bias = vo.to_f32(L.bias, dt).astype(np.float64) if L.bias is not None else 0.0
        ya = (q.astype(np.float64) + rs.astype(np.float64)) @ sx + bx + bias
        yb = vo.round_to(q + rs, dt).astype(np.float64) @ sx + bx + bias
        O_ = L.out_features
        ga = vo.from_f32(ya[:O_].astype(np.float32).reshape(1, 1, -1), dt)
        gb = vo.from_f32(yb[:O_].astype(np.float32).reshape(1, 1, -1), dt)
        ea.append(rel_err(ga, want, dt)); eb.append(rel_err(gb, want, dt))
    ea, eb = np.array(ea), np.array(eb)
    print(f"{dt}: folded (c + r exact)   max {ea.max():.2e} mean {ea.mean():.2e}   | folded with {dt}(c + r) first  max {eb.max():.2e} mean {eb.mean():.2e}   (bar {tol:g})")
