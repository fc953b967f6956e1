"""The C oracle (cpu_baseline "port") against the numpy oracle and the goldens."""
import hashlib

import numpy as np
import pytest

from oracle import vptq_oracle as vo
from oracle import c_oracle as co
from _cases import golden_names, load_golden, rel_err

pytestmark = pytest.mark.skipif(not co.available(), reason="run __graft_entry__.build() first")


def test_software_half_conversions_exhaustive():
    l = co.lib()
    allh = np.arange(65536, dtype=np.uint16)
    f = allh.view(np.float16).astype(np.float32)
    for h in range(0, 65536, 7):           # f16 -> f32 exact (NaN payloads aside)
        if not np.isnan(f[h]):
            assert l.vo_f16_to_f32(h) == f[h]
    rng = np.random.default_rng(0)
    xs = np.concatenate([rng.standard_normal(20000).astype(np.float32) * s
                         for s in (1e-8, 1e-6, 1e-4, 1, 100, 7e4)])
    xs = np.concatenate([xs, f[~np.isnan(f)], np.float32([65519.9, 65520.0, 2**-25, 2**-25 * 1.0001,
                                                           6.1e-5, 6.097e-5, -0.0, 0.0])])
    want_h = xs.astype(np.float16).view(np.uint16)
    want_b = vo.from_f32(xs, "bf16")
    for x, wh, wb in zip(xs[::13], want_h[::13], want_b[::13]):
        assert l.vo_f32_to_f16(float(x)) == wh, x
        assert l.vo_f32_to_bf16(float(x)) == wb, x


@pytest.mark.parametrize("name", golden_names())
def test_c_oracle_matches_reference_golden(name):
    L, x, y, cfg, W_head = load_golden(name)
    W = co.dequant(L)
    assert hashlib.sha256(W.tobytes()).hexdigest() == cfg["W_sha256"]
    out = co.forward(L, x)
    assert rel_err(out, y, cfg["dtype"]) <= (1e-3 if cfg["dtype"] == "f16" else 8e-3)
    assert (out == vo.forward(L, x)).mean() > 0.98
