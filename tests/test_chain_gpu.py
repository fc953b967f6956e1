"""GPU parity of the persistent layer-chain launch (`vptq_quant_gemv_chain`, gemv_k256c.hip):
every layer of a chain against the oracle / the reference goldens, through the C ABI.

Bar as everywhere: max|d| / max|ref| <= 1e-3 (fp16), 8e-3 (bf16)."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import vptq_oracle as vo
from _cases import rel_err, big_names, load_big, bit_identical_frac
from _gpu_util import spec_to_module, bits_to_tensor, tensor_to_bits, gemv_abi

pytestmark = pytest.mark.gpu
TOL = {"f16": 1e-3, "bf16": 8e-3}
CHAIN = 8   # VPTQ_GEMV_FORCE_MFMA: the chain kernel even where the layers are too small to fill the device


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from vptq_amd import _backend as B
    B.lib()
    return torch.device("cuda", 0)


def _x(I, dt, dist, seed, tokens=1):
    rng = np.random.default_rng(seed)
    xs = (0.02 + 0.5 * rng.standard_normal((1, tokens, I))) if dist == "ref-test" else rng.standard_normal((1, tokens, I))
    return vo.from_f32(xs.astype(np.float32), dt)


# (I, O, kwargs): shapes that hit every edge of the kernel's stream: rows not a multiple of 4
# (partial row group), columns not a multiple of 2048 (partial sweep), one and several sweeps,
# fewer / more row groups than workgroups, output bias
SHAPES = [
    (1024, 512, dict()),
    (4096, 264, dict()),
    (4104, 64, dict(bias=True)),
    (8192 + 512, 40, dict(dist="llm")),
    (512, 1000, dict(dist="llm", bias=True)),
    (2048, 2048 * 5, dict(dist="llm")),      # 1280 row groups: several per workgroup
    (6144, 24, dict()),
    (256, 8, dict()),                        # one row group, one partial sweep
]


def _build(shapes, dt, dev):
    Ls, ms, xs = [], [], []
    for i, (I, O, kw) in enumerate(shapes):
        kw = dict(kw)
        dist = kw.pop("dist", "ref-test")
        L = vo.make_layer(I, O, dist=dist, seed=1000 + 7 * i + I + O, dtype=dt, **kw)
        Ls.append(L)
        ms.append(spec_to_module(L, dev))
        xs.append(_x(I, dt, dist, 31 + i))
    return Ls, ms, xs


@pytest.mark.parametrize("dt", ["f16", "bf16"])
def test_independent_chain_every_layer_vs_oracle(dt, dev):
    from vptq_amd.ops.chain import GemvChain
    Ls, ms, xs = _build(SHAPES, dt, dev)
    chain = GemvChain(ms)
    # (the default arithmetic - the reference's roundings - runs inside the chain launch; bf16 since the end of round 6)
    assert chain.kernel_name(1, CHAIN) == "gemv_k256c_kernel"
    xt = [bits_to_tensor(x, dt, dev).reshape(x.shape) for x in xs]
    ys = chain(xt, flags=CHAIN)
    torch.cuda.synchronize()
    for L, x, y in zip(Ls, xs, ys):
        want = vo.forward(L, x)
        err = rel_err(tensor_to_bits(y), want, dt)
        assert err <= TOL[dt], f"{L.in_features}x{L.out_features}: {err:.3e}"
    # the same layers in another order and as single-layer chains: same bits (every layer's sums are
    # formed in the same order wherever it sits in the stream)
    order = [3, 0, 5, 7, 1, 6, 2, 4]
    ys2 = GemvChain([ms[i] for i in order])([xt[i] for i in order], flags=CHAIN)
    for j, i in enumerate(order):
        assert torch.equal(ys2[j].view(torch.int16), ys[i].view(torch.int16))
    for i in (0, 2, 5):
        y1 = GemvChain([ms[i]])([xt[i]], flags=CHAIN)[0]
        assert torch.equal(y1.view(torch.int16), ys[i].view(torch.int16))


@pytest.mark.parametrize("dt", ["f16", "bf16"])
def test_independent_chain_folded(dt, dev, folded_arithmetic):
    """the opt-in folded arithmetic through the chain launch, both dtypes: every layer against the oracle"""
    from vptq_amd.ops.chain import GemvChain
    Ls, ms, xs = _build(SHAPES, dt, dev)
    chain = GemvChain(ms)
    assert chain.kernel_name(1, CHAIN) == "gemv_k256c_kernel"
    xt = [bits_to_tensor(x, dt, dev).reshape(x.shape) for x in xs]
    ys = chain(xt, flags=CHAIN)
    for L, x, y in zip(Ls, xs, ys):
        err = rel_err(tensor_to_bits(y), vo.forward(L, x), dt)
        assert err <= TOL[dt], f"{L.in_features}x{L.out_features}: {err:.3e}"


def test_chain_float32_outputs(dev):
    from vptq_amd.ops.chain import GemvChain
    from vptq_amd import _backend as B
    Ls, ms, xs = _build(SHAPES[:4], "f16", dev)
    xt = [bits_to_tensor(x, "f16", dev).reshape(x.shape) for x in xs]
    y16 = GemvChain(ms)(xt, flags=CHAIN)
    y32 = GemvChain(ms)(xt, flags=B.GEMV_OUT_F32 | CHAIN)
    for a, b in zip(y16, y32):
        assert b.dtype == torch.float32
        assert torch.equal(b.half().view(torch.int16), a.view(torch.int16))   # one rounding of the same sums


@pytest.mark.parametrize("dt", ["f16", "bf16"])
def test_dependent_chain(dt, dev, folded_arithmetic):
    """x of layer i + 1 is y of layer i: every layer against the oracle applied to the GPU's own
    previous output (so the per-layer bar applies), and against the same layers launched one by one.  (The persistent
    dependent walk exists for the opt-in folded arithmetic only; with the default arithmetic a dependent chain is one launch
    per layer: test_dependent_chain_reference_arithmetic.)"""
    from vptq_amd.ops.chain import GemvChain
    dims = [1024, 2048, 512, 4096, 264, 1032, 1024]
    shapes = [(dims[i], dims[i + 1], dict(dist="llm", bias=(i % 3 == 1))) for i in range(len(dims) - 1)]
    Ls, ms, _ = _build(shapes, dt, dev)
    x0 = _x(dims[0], dt, "llm", 5)
    xt = bits_to_tensor(x0, dt, dev).reshape(x0.shape)
    chain = GemvChain(ms, dependent=True)
    assert chain.kernel_name(1, CHAIN) == "gemv_k256c_kernel"
    for rep in range(3):   # (the arrival counters are cleared by every call)
        ys = chain([xt], flags=CHAIN)
        torch.cuda.synchronize()
        xin = x0
        for L, y in zip(Ls, ys):
            want = vo.forward(L, xin)
            err = rel_err(tensor_to_bits(y), want, dt)
            assert err <= TOL[dt], f"rep {rep} {L.in_features}x{L.out_features}: {err:.3e}"
            xin = tensor_to_bits(y).reshape(1, 1, -1)
    # layer by layer through single-layer chains: identical bits
    xi = xt
    for m, y in zip(ms, ys):
        y1 = GemvChain([m])([xi], flags=CHAIN)[0]
        assert torch.equal(y1.view(torch.int16), y.view(torch.int16))
        xi = y1


def test_dependent_chain_reference_arithmetic(dev):
    """the product default: a dependent chain of layers in the reference's roundings = one launch per layer in stream
    order, the same bits as the modules called one after the other"""
    from vptq_amd.ops.chain import GemvChain
    dims = [1024, 2048, 512, 1024]
    shapes = [(dims[i], dims[i + 1], dict(dist="llm", bias=(i == 1))) for i in range(len(dims) - 1)]
    Ls, ms, _ = _build(shapes, "f16", dev)
    x0 = _x(dims[0], "f16", "llm", 5)
    xt = bits_to_tensor(x0, "f16", dev).reshape(x0.shape)
    chain = GemvChain(ms, dependent=True)
    assert chain.kernel_name(1) != "gemv_k256c_kernel"
    ys = chain([xt])
    xi, xin = xt, x0
    for L, m, y in zip(Ls, ms, ys):
        assert torch.equal(m(xi).view(torch.int16), y.view(torch.int16))
        assert rel_err(tensor_to_bits(y), vo.forward(L, xin), "f16") <= TOL["f16"]
        xi, xin = y, tensor_to_bits(y).reshape(1, 1, -1)


@pytest.mark.parametrize("arith", ["reference", "folded"])
@pytest.mark.parametrize("name", big_names())
def test_chain_on_reference_goldens_at_baseline_sizes(name, arith, dev, request):
    """hidden 4096 / 8192 layers whose y comes from the real reference (tests/golden/gen_golden_big.py):
    one-token, permutation-free cases through the chain launch, alone and 6 times in a row."""
    from vptq_amd.ops.chain import GemvChain
    L, x, y, cfg, _ = load_big(name)
    if cfg["tokens"] != 1 or cfg["perm"]:
        pytest.skip("the chain kernel takes one token, no permutation")
    dt = cfg["dtype"]
    if arith == "folded":
        request.getfixturevalue("folded_arithmetic")
    m = spec_to_module(L, dev)
    xt = bits_to_tensor(x, dt, dev).reshape(x.shape)
    chain = GemvChain([m] * 6)
    # (the reference's roundings run inside the chain launch, bf16 layers included since the end of round 6)
    assert chain.kernel_name(1, CHAIN) == "gemv_k256c_kernel"
    ys = chain([xt] * 6, flags=CHAIN)
    torch.cuda.synchronize()
    err = rel_err(tensor_to_bits(ys[0]), y, dt)
    assert err <= TOL[dt], f"{name}: {err:.3e}"
    for o in ys[1:]:
        assert torch.equal(o.view(torch.int16), ys[0].view(torch.int16))
    # against the library's one-launch-per-layer default (another kernel, same arithmetic form)
    ref = gemv_abi(m, xt, 0)
    assert rel_err(tensor_to_bits(ys[0]), tensor_to_bits(ref), dt) <= TOL[dt]


def test_chain_falls_back_layer_by_layer(dev):
    """a chain the one-launch kernel does not take (another format in it, 2 tokens) is executed as a grouped launch /
    single launches by the library: same results as the modules' forward"""
    from vptq_amd.ops.chain import GemvChain
    La = vo.make_layer(1024, 256, seed=3)
    Lb = vo.make_layer(1024, 256, seed=4, num_centroids=4096, num_res_centroids=0, vector_len=6)
    ma, mb = spec_to_module(La, dev), spec_to_module(Lb, dev)
    xa = bits_to_tensor(_x(1024, "f16", "ref-test", 1), "f16", dev).reshape(1, 1, 1024)
    chain = GemvChain([ma, mb])
    assert chain.kernel_name(1, 0) == "grouped"      # (vptq_quant_gemv_grouped serves mixed members one by one)
    ya, yb = chain([xa, xa])
    assert torch.equal(ya, ma(xa)) and torch.equal(yb, mb(xa))
    x2 = bits_to_tensor(_x(1024, "f16", "ref-test", 2, tokens=2), "f16", dev).reshape(1, 2, 1024)
    c2 = GemvChain([ma, ma])
    assert c2.kernel_name(2, 0) == "grouped"
    y2 = c2([x2, x2])
    assert torch.equal(y2[0], ma(x2)) and torch.equal(y2[1], ma(x2))


def test_chain_in_a_hipgraph_and_long_chains(dev):
    """40 layers = two launches; captured once, replayed with new activations"""
    from vptq_amd.ops.chain import GemvChain
    L = vo.make_layer(2048, 1024, seed=11, dist="llm")
    m = spec_to_module(L, dev)
    n = 40
    xs = [torch.zeros(1, 1, 2048, dtype=torch.float16, device=dev) for _ in range(n)]
    ys = [torch.empty(1, 1, 1024, dtype=torch.float16, device=dev) for _ in range(n)]
    chain = GemvChain([m] * n)
    chain(xs, ys, flags=CHAIN)          # warm-up outside the capture
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            chain(xs, ys, flags=CHAIN)
    for rep in range(2):
        vals = [_x(2048, "f16", "llm", 100 + rep * n + i) for i in range(n)]
        for xt, v in zip(xs, vals):
            xt.copy_(bits_to_tensor(v, "f16", dev).reshape(1, 1, 2048))
        g.replay()
        torch.cuda.synchronize()
        for i in (0, 17, 31, 32, 39):
            err = rel_err(tensor_to_bits(ys[i]), vo.forward(L, vals[i]), "f16")
            assert err <= 1e-3, (rep, i, err)


@pytest.mark.parametrize("dt", ["f16", "bf16"])
def test_chain_with_the_reference_roundings(dt, dev):
    """VPTQ_GEMV_EXACT inside the chain launch (independent layers; bf16 since the end of round 6: its roundings on the
    matrix pipe): every weight rebuilt with the reference CPU path's three roundings (vptq/ops/quant_gemm.py:143-158), so
    the outputs are the oracle's up to the order of the fp32 sums - over every edge shape of the stream; dependent chains
    take the per-layer kernels."""
    from vptq_amd.ops.chain import GemvChain
    from vptq_amd import _backend as B
    tol = 1e-3 if dt == "f16" else 8e-3
    Ls, ms, xs = _build(SHAPES, dt, dev)
    chain = GemvChain(ms)
    assert chain.kernel_name(1, CHAIN | B.GEMV_EXACT) == "gemv_k256c_kernel"
    xt = [bits_to_tensor(x, dt, dev).reshape(x.shape) for x in xs]
    ys = chain(xt, flags=CHAIN | B.GEMV_EXACT)
    torch.cuda.synchronize()
    for L, m, x, xg, y in zip(Ls, ms, xs, xt, ys):
        want = vo.forward(L, x)
        assert rel_err(tensor_to_bits(y), want, dt) <= tol, f"{L.in_features}x{L.out_features}"
        assert bit_identical_frac(tensor_to_bits(y), want) >= 0.95, f"{L.in_features}x{L.out_features}"
        # the per-layer kernel with the same roundings: the same weights, another order of the sums
        assert bit_identical_frac(tensor_to_bits(y), tensor_to_bits(gemv_abi(m, xg, B.GEMV_EXACT))) >= 0.95
    y32 = chain(xt, flags=CHAIN | B.GEMV_EXACT | B.GEMV_OUT_F32)
    for a, b in zip(ys, y32):
        assert torch.equal(b.to(a.dtype).view(torch.int16), a.view(torch.int16))


@pytest.mark.parametrize("exact", [False, True])
def test_chain_of_32_distinct_8192_layers_every_output(exact, dev):
    """>= 16 DISTINCT BASELINE-size layers through ONE launch, every layer's output checked (the ring of
    `bench.py`): 32 procedural 8192 x 8192 layers (tests/golden/_proc.py:big_tensors, one seed each) against the C
    oracle; default arithmetic within the bar, reference roundings >= 99 % bit-identical."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from _proc import big_tensors
    from oracle import c_oracle as co
    from vptq_amd.ops.chain import GemvChain
    from vptq_amd import _backend as B
    H, n = 8192, 32
    Ls, ms, xs = [], [], []
    for i in range(n):
        t = big_tensors(H, H, 8, 256, 256, False, False, 1, "f16", "llm" if i % 4 else "ref-test", 9100 + i)
        L = vo.LayerSpec(H, H, 8, 256, 256, 1, H, 0, -1, -1, "f16")
        L.indices = t["indices"]
        L.centroids, L.res_centroids = t["centroids"].reshape(1, 256, 8), t["res_centroids"].reshape(1, 256, 8)
        L.weight_scale, L.weight_bias = t["weight_scale"], t["weight_bias"]
        Ls.append(L); ms.append(spec_to_module(L, dev)); xs.append(t["x"].reshape(1, 1, H))
    chain = GemvChain(ms)
    flags = B.GEMV_EXACT if exact else 0
    assert chain.kernel_name(1, flags) == "gemv_k256c_kernel"   # (fills the device by itself: no FORCE flag)
    ys = chain([bits_to_tensor(x, "f16", dev).reshape(1, 1, H) for x in xs], flags=flags)
    torch.cuda.synchronize()
    worst = 0.0
    for i, (L, x, y) in enumerate(zip(Ls, xs, ys)):
        want = co.forward(L, x, quirk=False)
        err = rel_err(tensor_to_bits(y), want, "f16")
        worst = max(worst, err)
        assert err <= 1e-3, f"layer {i}: {err:.3e}"
        if exact:
            assert bit_identical_frac(tensor_to_bits(y), want) >= 0.99, f"layer {i}"
    print(f"32 distinct 8192^2 layers, exact={exact}: worst {worst:.2e}")


def test_chain_takes_layers_with_an_input_permutation(dev):
    """un-absorbed permutations must not drop a list to one launch per layer.  With the workspace
    `vptq_quant_gemv_chain_workspace_bytes_for` asks for (GemvChain provides it) x[perm] is gathered by one small launch in
    front of the persistent launch, which then runs on (x[perm], scale_permuted, bias_permuted); results = the layers' own
    forward.  Incl. one token of the reference's golden with perm + bias at hidden 4096."""
    from vptq_amd.ops.chain import GemvChain
    shapes = [(1024, 512, dict(dist="llm", enable_perm=True)), (4096, 264, dict(enable_perm=True, bias=True)),
              (2048, 2048, dict(dist="llm")), (4104, 64, dict(enable_perm=True)), (512, 1000, dict(dist="llm", enable_perm=True, bias=True))]
    Ls, ms, xs = _build(shapes, "f16", dev)
    chain = GemvChain(ms)
    assert chain.kernel_name(1, CHAIN) == "gemv_k256c_kernel"
    xt = [bits_to_tensor(x, "f16", dev).reshape(x.shape) for x in xs]
    ys = chain(xt, flags=CHAIN)
    torch.cuda.synchronize()
    for L, m, x, xg, y in zip(Ls, ms, xs, xt, ys):
        assert rel_err(tensor_to_bits(y), vo.forward(L, x), "f16") <= 1e-3, f"{L.in_features}x{L.out_features}"
        assert rel_err(tensor_to_bits(y), tensor_to_bits(gemv_abi(m, xg, 0)), "f16") <= 1e-3
    ys2 = chain(xt, flags=CHAIN)      # (the workspace is reused: same bits)
    for a, b in zip(ys, ys2):
        assert torch.equal(a.view(torch.int16), b.view(torch.int16))
    # the reference's golden with perm + bias (2 tokens in the fixture: one of them here), 8 times in one launch
    L, x, y, cfg, _ = load_big("h4096_f16_llm_perm_bias_t2")
    m = spec_to_module(L, dev)
    x0 = bits_to_tensor(x[:, :1], "f16", dev).reshape(1, 1, -1)
    c8 = GemvChain([m] * 9)
    assert c8.kernel_name(1, 0) == "gemv_k256c_kernel"     # (9 layers of 4096^2 fill the device: no FORCE flag)
    out = c8([x0] * 9)
    torch.cuda.synchronize()
    assert rel_err(tensor_to_bits(out[0]), y[:, :1], "f16") <= 1e-3
    for o in out[1:]:
        assert torch.equal(o.view(torch.int16), out[0].view(torch.int16))
    # in a hipGraph
    g, s = torch.cuda.CUDAGraph(), torch.cuda.Stream()
    with torch.cuda.stream(s):
        chain(xt, flags=CHAIN)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            yg = chain(xt, flags=CHAIN)
    g.replay()
    torch.cuda.synchronize()
    for a, b in zip(ys, yg):
        assert torch.equal(a.view(torch.int16), b.view(torch.int16))


# ---- VPTQ_GEMV_SELECTIVE (round 6): folded form + the reference's roundings on the blocks of columns an activation dominates ----
def _spiky(x_bits, dt, cols, gain):
    """activation bits with `cols` columns multiplied by `gain` (massive-activation channels)"""
    x = vo.to_f32(x_bits, dt).copy()
    x[..., cols] *= gain
    return vo.from_f32(x.astype(np.float32), dt)


def test_selective_chain_vs_oracle(dev, selective_arithmetic):
    """every edge shape of the stream in the selective arithmetic, dense activations and activations with massive channels
    (which the folded form alone does not survive: profiles/r05/gate_count_*_folded_opt_in.txt): within the bar of the oracle"""
    from vptq_amd.ops.chain import GemvChain
    from vptq_amd import _backend as B
    Ls, ms, xs = _build(SHAPES, "f16", dev)
    chain = GemvChain(ms)
    assert chain.kernel_name(1, CHAIN) == "gemv_k256c_kernel"
    for variant in ("dense", "massive"):
        xv = xs if variant == "dense" else [_spiky(x, "f16", [3, x.shape[-1] // 2 + 5, x.shape[-1] - 1], 40.0) for x in xs]
        xt = [bits_to_tensor(x, "f16", dev).reshape(x.shape) for x in xv]
        ys = chain(xt, flags=CHAIN)
        torch.cuda.synchronize()
        for L, m, x, y in zip(Ls, ms, xv, ys):
            flag = m._descriptor()[9]
            assert flag in (B.GEMV_SELECTIVE, B.GEMV_EXACT)   # (EXACT: the load-time gate refused the layer)
            err = rel_err(tensor_to_bits(y), vo.forward(L, x), "f16")
            assert err <= 1e-3, f"{variant} {L.in_features}x{L.out_features}: {err:.3e}"


def test_selective_hot_blocks_take_the_reference_roundings(dev, folded_arithmetic):
    """activations that are ZERO outside a few massive columns: every column that contributes sits in a hot block, so the
    selective launch must reproduce the launch with VPTQ_GEMV_EXACT (same weights bit for bit; fp32 sums of <= 3 terms in another
    order) - and the folded launch of the same inputs must NOT (else the test proves nothing)"""
    from vptq_amd.ops.chain import GemvChain
    from vptq_amd import _backend as B
    Ls, ms, xs = _build(SHAPES[:6], "f16", dev)
    xt = []
    for x in xs:
        I = x.shape[-1]
        xf = np.zeros((1, 1, I), dtype=np.float32)
        for c, v in ((1, 23.5), (I // 3, -17.25), (I - 2, 9.125)):
            xf[0, 0, c] = v
        xt.append(bits_to_tensor(vo.from_f32(xf, "f16"), "f16", dev).reshape(1, 1, I))
    chain = GemvChain(ms)
    y_sel = chain(xt, flags=CHAIN | B.GEMV_SELECTIVE | B.GEMV_OUT_F32)
    y_ex = chain(xt, flags=CHAIN | B.GEMV_EXACT | B.GEMV_OUT_F32)
    y_fo = chain(xt, flags=CHAIN | B.GEMV_OUT_F32)
    torch.cuda.synchronize()
    differs = 0
    for L, a, b, c in zip(Ls, y_sel, y_ex, y_fo):
        den = float(b.abs().max())
        assert float((a - b).abs().max()) <= 2e-6 * den, f"{L.in_features}x{L.out_features}"   # (fp32 summation order only)
        differs += int(float((c - b).abs().max()) > 1e-5 * den)
    assert differs >= 4, differs
    # ... and rounded: bit-identical to the exact launch
    for a, b in zip(chain(xt, flags=CHAIN | B.GEMV_SELECTIVE), chain(xt, flags=CHAIN | B.GEMV_EXACT)):
        assert float((a.view(torch.int16) == b.view(torch.int16)).float().mean()) >= 0.999


def test_selective_dense_activations_run_the_folded_loop(dev, folded_arithmetic):
    """without a dominant column nothing is hot: the selective launch IS the folded launch, bit for bit"""
    from vptq_amd.ops.chain import GemvChain
    from vptq_amd import _backend as B
    Ls, ms, _ = _build(SHAPES[:6], "f16", dev)
    # +-1 activations: |f16(s x)| = |s|, and the layers' scales stay far below 6 x their rms
    xt = []
    for i, L in enumerate(Ls):
        rng = np.random.default_rng(77 + i)
        xf = np.where(rng.random((1, 1, L.in_features)) < 0.5, -1.0, 1.0).astype(np.float32)
        xt.append(bits_to_tensor(vo.from_f32(xf, "f16"), "f16", dev).reshape(xf.shape))
    for m in ms:
        s = m.weight_scale.float().abs()
        assert float(s.max()) < 5.5 * float(s.pow(2).mean().sqrt()), "pick another seed: a scale outlier makes a column hot"
    chain = GemvChain(ms)
    for a, b in zip(chain(xt, flags=CHAIN | B.GEMV_SELECTIVE), chain(xt, flags=CHAIN)):
        assert torch.equal(a.view(torch.int16), b.view(torch.int16))


def test_selective_without_workspace_or_kernel_is_exact(dev, folded_arithmetic):
    """VPTQ_GEMV_SELECTIVE where nothing implements it - no workspace for the chain's thresholds, the one-layer entry point, bf16 -
    is VPTQ_GEMV_EXACT (always at least as close to the reference); EXACT wins when both are set"""
    from vptq_amd import _backend as B
    Ls, ms, xs = _build(SHAPES[:3], "f16", dev)
    xt = [bits_to_tensor(x, "f16", dev).reshape(x.shape) for x in xs]
    n = len(ms)
    descs = (B.LayerDesc * n)(*[m._descriptor()[1] for m in ms])
    xp, yp = (C.c_void_p * n)(), (C.c_void_p * n)()

    def run(flags, ws=None):
        ys = [torch.empty(1, 1, m.out_features, dtype=torch.float16, device=dev) for m in ms]
        for i in range(n):
            xp[i], yp[i] = xt[i].data_ptr(), ys[i].data_ptr()
        rc = B.lib().vptq_quant_gemv_chain(descs, n, xp, yp, 1, flags, None if ws is None else ws.data_ptr(),
                                           0 if ws is None else ws.numel(), B.current_stream_ptr(dev))
        assert rc == 0, B.lib().vptq_last_error()
        torch.cuda.synchronize()
        return ys
    want = run(CHAIN | B.GEMV_EXACT)
    for got in (run(CHAIN | B.GEMV_SELECTIVE), run(CHAIN | B.GEMV_SELECTIVE | B.GEMV_EXACT)):
        for a, b in zip(got, want):
            assert torch.equal(a.view(torch.int16), b.view(torch.int16))
    need = B.lib().vptq_quant_gemv_chain_workspace_bytes_for(descs, n, B.GEMV_SELECTIVE)
    assert need >= 256 + sum(4 * 8 * ((m.out_features + 7) // 8) for m in ms)
    assert B.lib().vptq_quant_gemv_chain_workspace_bytes_for(descs, n, B.GEMV_SELECTIVE | B.GEMV_EXACT) == 0
    ws = torch.zeros(need, dtype=torch.uint8, device=dev)
    sel = run(CHAIN | B.GEMV_SELECTIVE, ws)
    for L, x, y in zip(Ls, xs, sel):
        assert rel_err(tensor_to_bits(y), vo.forward(L, x), "f16") <= 1e-3
    # the one-layer entry point
    for m, x, w in zip(ms, xt, want):
        a = gemv_abi(m, x, B.GEMV_SELECTIVE)
        b = gemv_abi(m, x, B.GEMV_EXACT)
        assert torch.equal(a.view(torch.int16), b.view(torch.int16))
