"""VPTQ_GEMV_SELECTIVE through the one-layer / grouped entry points (round 6): the persistent MFMA kernel of the canonical format in
the folded form, with the reference's roundings on the blocks of 128 columns an activation dominates - decided, zeroed and
corrected INSIDE the launch (gemv_k256m.hip).  Through the C ABI, against the oracle and against the two pure arithmetics.

Bar as everywhere: max|d| / max|ref| <= 1e-3 (fp16)."""
import numpy as np
import pytest
import torch

from oracle import vptq_oracle as vo
from _cases import rel_err
from _gpu_util import spec_to_module, bits_to_tensor, tensor_to_bits, gemv_abi, kernel_name

pytestmark = pytest.mark.gpu
MFMA = 1 << 3   # VPTQ_GEMV_FORCE_MFMA: the persistent kernel on layers too small to take it by themselves


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from vptq_amd import _backend as B
    B.lib()
    return torch.device("cuda", 0)


def _x(I, seed, dist="llm"):
    rng = np.random.default_rng(seed)
    xs = (0.02 + 0.5 * rng.standard_normal((1, 1, I))) if dist == "ref-test" else rng.standard_normal((1, 1, I))
    return xs.astype(np.float32)


def _col_scale(L):
    """weight_scale in COLUMN order (column c of the quantised matrix multiplies input feature perm[c])"""
    s = vo.to_f32(np.asarray(L.weight_scale), "f16").reshape(-1)
    return s


# (I, O, kwargs): one to seven sweeps, partial sweeps and row groups, several row groups per workgroup, bias, permutation
SHAPES = [
    (2048, 4608, dict(dist="llm")),                       # 144 row groups: the kernel's own threshold
    (8192, 512, dict(dist="llm", bias=True)),
    (4104, 264, dict()),
    (1024, 2048 * 3, dict(dist="llm")),                   # 768 row groups: 3 per workgroup
    (6144, 40, dict(dist="llm", enable_perm=True)),
    (14336, 72, dict(dist="llm", bias=True)),             # 7 sweeps, staged in two chunks
    (256, 8, dict()),
]


@pytest.mark.parametrize("I,O,kw", SHAPES)
def test_selective_one_layer_vs_oracle_and_pure_arithmetics(I, O, kw, dev):
    from vptq_amd import _backend as B
    kw = dict(kw)
    dist = kw.pop("dist", "ref-test")
    L = vo.make_layer(I, O, dist=dist, seed=3 * I + O, dtype="f16", **kw)
    m = spec_to_module(L, dev)
    assert kernel_name(m, 1, B.GEMV_SELECTIVE | MFMA) == "gemv_k256m_kernel<selective>"
    assert kernel_name(m, 1, B.GEMV_SELECTIVE | B.GEMV_EXACT | MFMA) in ("gemv_k256m_kernel", "gemv_k256_kernel")   # EXACT wins (7 sweeps: the VALU kernel)
    assert kernel_name(m, 2, B.GEMV_SELECTIVE | MFMA) in ("gemv_k256m_kernel", "gemv_k256_kernel")   # one token only: the reference's roundings
    xd = _x(I, I + 1, dist)
    # dense, massive channels, and ONLY massive channels (every contributing column sits in a hot block)
    xm = xd.copy()
    hot_cols = [1, I // 3, I - 2]
    xm[..., hot_cols] *= 40.0
    xo = np.zeros_like(xd)
    sc = _col_scale(L)
    for c, v in zip(hot_cols, (23.5, -17.25, 9.125)):
        xo[..., c] = v / max(abs(float(sc[c])), 1e-3)     # (|s x| of comparable size: all three columns are hot whatever their scales)
    for name, xf in (("dense", xd), ("massive", xm), ("only-massive", xo)):
        xb = vo.from_f32(xf, "f16")
        xt = bits_to_tensor(xb, "f16", dev).reshape(1, 1, I)
        y = gemv_abi(m, xt, B.GEMV_SELECTIVE | MFMA)
        torch.cuda.synchronize()
        err = rel_err(tensor_to_bits(y), vo.forward(L, xb), "f16")
        assert err <= 1e-3, f"{name} {I}x{O}: {err:.3e}"
        ys = gemv_abi(m, xt, B.GEMV_SELECTIVE | MFMA, out_f32=True)
        ye = gemv_abi(m, xt, B.GEMV_EXACT | MFMA, out_f32=True)
        yf = gemv_abi(m, xt, MFMA, out_f32=True)
        den = float(ye.abs().max())
        if name == "only-massive":
            # the corrections ARE the exact launch's sums (fp32 summation order apart), and the folded form is somewhere else
            assert float((ys - ye).abs().max()) <= 2e-6 * den, f"{I}x{O}"
        if name == "massive":
            # closer to the reference's roundings than the folded form wherever that one is off at all
            assert float((ys - ye).abs().max()) <= float((yf - ye).abs().max()) + 1e-7 * den
        # deterministic
        assert torch.equal(gemv_abi(m, xt, B.GEMV_SELECTIVE | MFMA).view(torch.int16), y.view(torch.int16))


def test_selective_without_a_hot_block_is_the_folded_launch(dev):
    """+-1 activations on a layer whose scales stay below 6 x their rms in every 512-column group: nothing is hot, same bits as <fast>"""
    from vptq_amd import _backend as B
    for I, O in ((2048, 4608), (8192, 512)):
        L = vo.make_layer(I, O, dist="llm", seed=I + O, dtype="f16")
        m = spec_to_module(L, dev)
        s = m.weight_scale.float().abs().view(-1, 512)
        assert bool((s.amax(1) < 5.5 * s.pow(2).mean(1).sqrt()).all())
        rng = np.random.default_rng(5)
        xf = np.where(rng.random((1, 1, I)) < 0.5, -1.0, 1.0).astype(np.float32)
        xt = bits_to_tensor(vo.from_f32(xf, "f16"), "f16", dev).reshape(1, 1, I)
        a = gemv_abi(m, xt, B.GEMV_SELECTIVE | MFMA)
        b = gemv_abi(m, xt, MFMA)
        assert torch.equal(a.view(torch.int16), b.view(torch.int16))


def test_selective_grouped_siblings_and_the_fallbacks(dev):
    """q / k / v-like siblings in one grouped launch; layers the kernel does not take in this mode get the reference's roundings"""
    import ctypes as C
    from vptq_amd import _backend as B
    I = 4096
    Ls = [vo.make_layer(I, O, dist="llm", seed=50 + O, dtype="f16") for O in (4096, 1024, 1024)]
    ms = [spec_to_module(L, dev) for L in Ls]
    xf = _x(I, 9)
    xf[..., [7, 2000, 4095]] *= 35.0
    xb = vo.from_f32(xf, "f16")
    xt = bits_to_tensor(xb, "f16", dev).reshape(1, 1, I)
    n = len(ms)
    descs = (B.LayerDesc * n)(*[m._descriptor()[1] for m in ms])
    assert B.lib().vptq_quant_gemv_grouped_kernel_name(descs, n, 1, B.GEMV_SELECTIVE).decode() == "gemv_k256m_kernel<selective>"
    ys = [torch.empty(1, 1, m.out_features, dtype=torch.float16, device=dev) for m in ms]
    xp = (C.c_void_p * n)(*[xt.data_ptr()] * n)
    yp = (C.c_void_p * n)(*[y.data_ptr() for y in ys])
    rc = B.lib().vptq_quant_gemv_grouped(descs, n, xp, yp, 1, B.GEMV_SELECTIVE, B.current_stream_ptr(dev))
    assert rc == 0, B.lib().vptq_last_error()
    torch.cuda.synchronize()
    for L, m, y in zip(Ls, ms, ys):
        assert rel_err(tensor_to_bits(y), vo.forward(L, xb), "f16") <= 1e-3
        # the same layer alone: the same arithmetic (its own launch: another share of the CUs, the same sums)
        assert torch.equal(gemv_abi(m, xt, B.GEMV_SELECTIVE | MFMA).view(torch.int16), y.view(torch.int16))
    # the small-layer VALU kernel and layers whose workgroups walk more than 4 row groups: the reference's roundings
    # (bf16 layers take the selective form like fp16 ones: test_selective_bf16_one_layer_and_chain)
    Lsmall = vo.make_layer(1024, 256, dist="llm", seed=2, dtype="f16")
    msmall = spec_to_module(Lsmall, dev)
    assert kernel_name(msmall, 1, B.GEMV_SELECTIVE) == "gemv_k256_kernel"
    xs = bits_to_tensor(vo.from_f32(_x(1024, 3), "f16"), "f16", dev).reshape(1, 1, 1024)
    assert torch.equal(gemv_abi(msmall, xs, B.GEMV_SELECTIVE).view(torch.int16), gemv_abi(msmall, xs, B.GEMV_EXACT).view(torch.int16))
    # 1280 row groups: 5 per workgroup - the corrections are built in two rounds of 4 row groups (they hold 16)
    Ltall = vo.make_layer(512, 8192 * 5, dist="llm", seed=4, dtype="f16")
    mtall = spec_to_module(Ltall, dev)
    assert kernel_name(mtall, 1, B.GEMV_SELECTIVE) == "gemv_k256m_kernel<selective>"
    xtl = _x(512, 6)
    xtl[..., [3, 300]] *= 40.0
    xtb = vo.from_f32(xtl, "f16")
    ytl = gemv_abi(mtall, bits_to_tensor(xtb, "f16", dev).reshape(1, 1, 512), B.GEMV_SELECTIVE)
    assert rel_err(tensor_to_bits(ytl), vo.forward(Ltall, xtb), "f16") <= 1e-3


def test_module_forward_in_the_selective_arithmetic(dev, selective_arithmetic):
    """vptq_amd.set_arithmetic("selective"): VQuantLinear.forward of a layer the persistent kernel takes; siblings; other formats
    (large codebooks: the exact sliced route / gather kernels) keep the reference's roundings"""
    import vptq_amd
    from vptq_amd import _backend as B
    assert vptq_amd.arithmetic() == "selective"
    L = vo.make_layer(8192, 8192, dist="llm", seed=11, dtype="f16")
    m = spec_to_module(L, dev)
    assert m._descriptor()[9] in (B.GEMV_SELECTIVE, B.GEMV_EXACT)
    xf = _x(8192, 12)
    xf[..., [100, 5000]] *= 50.0
    xb = vo.from_f32(xf, "f16")
    y = m(bits_to_tensor(xb, "f16", dev).reshape(1, 1, 8192))
    torch.cuda.synchronize()
    assert rel_err(tensor_to_bits(y), vo.forward(L, xb), "f16") <= 1e-3
    if m._descriptor()[9] == B.GEMV_SELECTIVE:
        assert kernel_name(m, 1, m._descriptor()[9]) == "gemv_k256m_kernel<selective>"
    L2 = vo.make_layer(2048, 264, dist="llm", seed=13, dtype="f16", num_centroids=65536, num_res_centroids=256)
    m2 = spec_to_module(L2, dev)
    x2 = vo.from_f32(_x(2048, 14), "f16")
    y2 = m2(bits_to_tensor(x2, "f16", dev).reshape(1, 1, 2048))
    want2 = vo.forward(L2, x2)
    assert rel_err(tensor_to_bits(y2), want2, "f16") <= 1e-3
    assert float((tensor_to_bits(y2).reshape(-1) == np.asarray(want2).reshape(-1)).mean()) >= 0.95   # (the reference's roundings)


# ---- the large-codebook formats: VPTQ_GEMV_SELECTIVE over the FOLDED sliced layouts (gemv_hot.hip + gemv_sliced.hip) ----
SLICED_CASES = [
    # (I, O, v, kr, kwargs)
    (4096, 1024, 8, 65536, dict(dist="llm")),
    (8192, 520, 8, 65536, dict(dist="llm", bias=True)),
    (2048, 1032, 8, 4096, dict(dist="llm", enable_perm=True)),
    (4104, 264, 8, 65536, dict()),
    (4096, 1024, 16, 65536, dict(dist="llm")),
    (2048, 528, 16, 65536, dict(dist="llm", enable_perm=True, bias=True)),
    (4096, 512, 8, 256, dict(dist="llm")),          # one table + the 256-entry residual (24-bit elements)
    (4096, 512, 8, 0, dict(dist="llm")),            # no residual codebook
]


@pytest.mark.parametrize("I,O,v,kr,kw", SLICED_CASES)
def test_selective_over_sliced_layouts(I, O, v, kr, kw, dev):
    from vptq_amd.utils.sliced import SlicedGemv
    from vptq_amd import _backend as B
    kw = dict(kw)
    dist = kw.pop("dist", "ref-test")
    L = vo.make_layer(I, O, dist=dist, seed=I + O + v + kr, dtype="f16", vector_len=v, num_centroids=65536, num_res_centroids=kr, **kw)
    m = spec_to_module(L, dev)
    assert B.lib().vptq_quant_gemv_sliced_selective_supported(m._descriptor()[1]) == 1
    sl = SlicedGemv(m, selective=True)
    fo = SlicedGemv(m)
    assert sl.selective and not sl.exact and not sl.tokens_supported(2)
    xd = _x(I, I + 7, dist)
    xm = xd.copy()
    hot_cols = [2, I // 2 + 1, I - 3]
    xm[..., hot_cols] *= 40.0
    xo = np.zeros_like(xd)
    sc = _col_scale(L)
    for c, val in zip(hot_cols, (21.5, -13.25, 7.125)):
        xo[..., c] = val / max(abs(float(sc[c])), 1e-3)   # (|s x| of comparable size: all three columns are hot whatever their scales)
    for name, xf in (("dense", xd), ("massive", xm), ("only-massive", xo)):
        xb = vo.from_f32(xf, "f16")
        xt = bits_to_tensor(xb, "f16", dev).reshape(1, 1, I)
        y = sl(xt)
        assert y is not None
        torch.cuda.synchronize()
        err = rel_err(tensor_to_bits(y), vo.forward(L, xb), "f16")
        assert err <= 1e-3, f"{name} {I}x{O} v{v} kr{kr}: {err:.3e}"
        ys = sl(xt, flags=B.GEMV_OUT_F32)
        ye = gemv_abi(m, xt, B.GEMV_EXACT, out_f32=True)         # the gather kernels: the reference's roundings
        yf = fo(xt, flags=B.GEMV_OUT_F32)
        den = float(ye.abs().max())
        if name == "only-massive":
            assert float((ys - ye).abs().max()) <= 3e-6 * den, f"{I}x{O} v{v} kr{kr}"
        if name == "massive":
            assert float((ys - ye).abs().max()) <= float((yf - ye).abs().max()) + 1e-6 * den
        assert torch.equal(sl(xt).view(torch.int16), y.view(torch.int16))     # deterministic; the workspace is left clean
    # without a hot block: the folded launch, bit for bit (+-1 activations against scales without outliers)
    s = m.weight_scale.float().abs()
    if float(s.max()) < 5.5 * float(s.pow(2).mean().sqrt()):
        rng = np.random.default_rng(3)
        xf = np.where(rng.random((1, 1, I)) < 0.5, -1.0, 1.0).astype(np.float32)
        xt = bits_to_tensor(vo.from_f32(xf, "f16"), "f16", dev).reshape(1, 1, I)
        assert torch.equal(sl(xt).view(torch.int16), fo(xt).view(torch.int16))


def test_module_takes_the_selective_sliced_route_for_two_table_formats(dev, selective_arithmetic):
    """set_arithmetic("selective"): a two-table large-codebook layer big enough for the sliced routes runs the folded layouts with the
    pre-pass; one-table formats keep the reference's roundings over their exact layouts"""
    from vptq_amd import _backend as B
    L = vo.make_layer(4096, 4096, dist="llm", seed=21, dtype="f16", num_centroids=65536, num_res_centroids=65536)
    m = spec_to_module(L, dev)
    xf = _x(4096, 22)
    xf[..., [9, 3000]] *= 45.0
    xb = vo.from_f32(xf, "f16")
    y = m(bits_to_tensor(xb, "f16", dev).reshape(1, 1, 4096))
    torch.cuda.synchronize()
    assert rel_err(tensor_to_bits(y), vo.forward(L, xb), "f16") <= 1e-3
    sl = m.__dict__.get("_sliced")
    assert sl is not None and sl[1] is not None
    if m._descriptor()[9] == B.GEMV_SELECTIVE:
        assert sl[1].selective or sl[1].exact        # (exact: the load-time gate refused the selective form of this layer)
    L1 = vo.make_layer(4096, 2048, dist="llm", seed=23, dtype="f16", num_centroids=65536, num_res_centroids=256)
    m1 = spec_to_module(L1, dev)
    x1 = vo.from_f32(_x(4096, 24), "f16")
    y1 = m1(bits_to_tensor(x1, "f16", dev).reshape(1, 1, 4096))
    w1 = vo.forward(L1, x1)
    assert rel_err(tensor_to_bits(y1), w1, "f16") <= 1e-3
    s1 = m1.__dict__.get("_sliced")
    assert s1 is not None and s1[1] is not None and s1[1].exact


# ---- bf16 (the dtype of most published checkpoints): the default arithmetic runs the widened VALU kernel there (21.5 us per 8192^2
# layer); the selective form = the dtype-agnostic folded MFMA loop + corrections in widened arithmetic on the hot blocks only
@pytest.mark.parametrize("I,O,kw", [SHAPES[0], SHAPES[1], SHAPES[4]])
def test_selective_bf16_one_layer_and_chain(I, O, kw, dev):
    from vptq_amd import _backend as B
    from vptq_amd.ops.chain import GemvChain
    import vptq_amd
    kw = dict(kw)
    dist = kw.pop("dist", "ref-test")
    L = vo.make_layer(I, O, dist=dist, seed=5 * I + O, dtype="bf16", **kw)
    m = spec_to_module(L, dev)
    assert kernel_name(m, 1, B.GEMV_SELECTIVE | MFMA) == "gemv_k256m_kernel<selective>"
    xd = _x(I, I + 2, dist)
    xm = xd.copy()
    xm[..., [1, I // 3, I - 2]] *= 40.0
    before = vptq_amd.arithmetic()
    vptq_amd.set_arithmetic("folded")      # (descriptor flags 0: the call's flags decide inside GemvChain)
    try:
        for name, xf in (("dense", xd), ("massive", xm)):
            xb = vo.from_f32(xf, "bf16")
            xt = bits_to_tensor(xb, "bf16", dev).reshape(1, 1, I)
            want = vo.forward(L, xb)
            y = gemv_abi(m, xt, B.GEMV_SELECTIVE | MFMA)
            torch.cuda.synchronize()
            assert rel_err(tensor_to_bits(y), want, "bf16") <= 8e-3, f"{name} {I}x{O}"
            if not kw.get("enable_perm"):
                ch = GemvChain([m, m])
                assert ch.kernel_name(1, B.GEMV_SELECTIVE | MFMA) == "gemv_k256c_kernel"
                for yc in ch([xt, xt], flags=B.GEMV_SELECTIVE | MFMA):
                    assert rel_err(tensor_to_bits(yc), want, "bf16") <= 8e-3, f"chain {name} {I}x{O}"
            if name == "massive":
                ys = gemv_abi(m, xt, B.GEMV_SELECTIVE | MFMA, out_f32=True)
                ye = gemv_abi(m, xt, B.GEMV_EXACT, out_f32=True)       # (bf16: the widened VALU kernel)
                yf = gemv_abi(m, xt, MFMA, out_f32=True)
                assert float((ys - ye).abs().max()) <= float((yf - ye).abs().max()) + 1e-6 * float(ye.abs().max())
    finally:
        vptq_amd.set_arithmetic(before)


def test_selective_bf16_over_sliced_layouts(dev):
    from vptq_amd.utils.sliced import SlicedGemv
    from vptq_amd import _backend as B
    for (I, O, v) in ((4096, 1024, 8), (2048, 528, 16)):
        L = vo.make_layer(I, O, dist="llm", seed=I + v, dtype="bf16", vector_len=v, num_centroids=65536, num_res_centroids=65536)
        m = spec_to_module(L, dev)
        sl = SlicedGemv(m, selective=True)
        xf = _x(I, 3)
        xf[..., [4, I // 2, I - 1]] *= 40.0
        xb = vo.from_f32(xf, "bf16")
        y = sl(bits_to_tensor(xb, "bf16", dev).reshape(1, 1, I))
        torch.cuda.synchronize()
        assert rel_err(tensor_to_bits(y), vo.forward(L, xb), "bf16") <= 8e-3
