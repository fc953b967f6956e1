#!/usr/bin/env python3
"""Randomised parity run of the canonical-format GEMV kernels against the numpy oracle (GPU box).

    python tools/gpu_fuzz.py [--cases 40] [--seed 0]

Random widths (multiples of 8 up to 30000), heights, permutation / output bias, every kernel
and arithmetic flag combination; prints the worst max-normalised error per flag set.
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

FLAGS = {"default": 0, "exact": 4, "mfma": 8, "mfma+exact": 12, "valu": 16, "valu+exact": 20}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=40)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--dtype", default="f16", choices=["f16", "bf16"])
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
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
