"""GPU parity of the one-token GEMV over the load-time derived sliced layout of v8-k65536-0 layers
(vptq_quant_gemv_sliced, gemv_sliced.hip; layout built by vptq_amd/utils/sliced.py): against the oracle, the
reference's goldens and the library's gather kernel, through the C ABI.

Bar as everywhere: max|d| / max|ref| <= 1e-3 (fp16), 8e-3 (bf16)."""
import numpy as np
import pytest
import torch

from oracle import vptq_oracle as vo
from _cases import rel_err, fmt_names, load_fmt, load_golden
from _gpu_util import spec_to_module, bits_to_tensor, tensor_to_bits, gemv_abi, kernel_name

pytestmark = pytest.mark.gpu
TOL = {"f16": 1e-3, "bf16": 8e-3}
EXACT = 1 << 2


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from vptq_amd import _backend as B
    B.lib()
    return torch.device("cuda", 0)


def _x(I, dt, dist, seed):
    rng = np.random.default_rng(seed)
    xs = (0.02 + 0.5 * rng.standard_normal((1, 1, I))) if dist == "ref-test" else rng.standard_normal((1, 1, I))
    return vo.from_f32(xs.astype(np.float32), dt)


# I, O, kwargs, rows per wave (0 = the builder's choice): rows not a multiple of the 16-wave block, several rows
# per wave, columns not a multiple of 64, output bias, a narrow layer (most slices of a row nearly empty)
CASES = [
    (1024, 256, dict(), 0),
    (2048, 1032, dict(bias=True), 0),
    (2048, 1032, dict(bias=True), 3),
    (4104, 264, dict(dist="llm"), 0),
    (512, 4096, dict(dist="llm"), 2),
    (1024, 4608, dict(dist="llm", bias=True), 18),   # more than 16 rows per wave: the lanes that hold the rows' words come round again
    (64, 72, dict(), 0),
    (8192, 512, dict(dist="llm", bias=True), 1),
    (2048, 520, dict(dist="llm", enable_perm=True, bias=True), 0),   # a permutation: applied while the activations are staged
    (6152, 264, dict(enable_perm=True), 2),
    (14336, 128, dict(dist="llm"), 0),          # with a residual codebook: 16 slices of 4096 entries
    (28672, 64, dict(dist="llm", bias=True), 0),  # 16 slices
]


@pytest.mark.parametrize("kr", [0, 256, 65536])
@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("I,O,kw,rpw", CASES)
def test_sliced_layout_gemv_vs_oracle(I, O, kw, rpw, dt, kr, dev):
    from vptq_amd.utils.sliced import SlicedGemv
    from vptq_amd import _backend as B
    kw = dict(kw)
    dist = kw.pop("dist", "ref-test")
    L = vo.make_layer(I, O, dist=dist, seed=I + O, dtype=dt, num_centroids=65536, num_res_centroids=kr, **kw)
    x = _x(I, dt, dist, I)
    m = spec_to_module(L, dev)
    assert kernel_name(m, 1) == "gemv_gather_kernel"
    sl = SlicedGemv(m, rows_per_wave=rpw)
    xt = bits_to_tensor(x, dt, dev).reshape(x.shape)
    got = sl(xt)
    torch.cuda.synchronize()
    want = vo.forward(L, x)
    err = rel_err(tensor_to_bits(got), want, dt)
    assert err <= TOL[dt], f"{I}x{O} {dt}: {err:.3e}"
    assert sl.slices == (8 if I <= (14080 if kr == 256 else 14336) else 16)   # (kr = 65536: one table per pass, nothing beside the slice)
    # against the library's own route for this layer (gather kernel, the reference's roundings)
    assert rel_err(tensor_to_bits(got), tensor_to_bits(gemv_abi(m, xt, 0)), dt) <= TOL[dt]
    # fp32 outputs: one rounding of the same sums; determinism
    y32 = sl(xt, flags=B.GEMV_OUT_F32)
    assert y32.dtype == torch.float32 and torch.equal(y32.to(got.dtype).view(torch.int16), got.view(torch.int16))
    assert torch.equal(sl(xt).view(torch.int16), got.view(torch.int16))
    # memory: 4 bytes per element + padding, on top of the packed indices
    # memory: 4 (5) bytes per element and layout + padding (< 64 elements per slice and row) + the block tables, on top of the packed indices
    n_lay = 2 if kr == 65536 else 1
    assert sl.extra_bytes <= 2.0 * m.indices.numel() * 4 + n_lay * sl.slices * (8 + 64 * 5) * m.indices.shape[1]


# ---- the reference's roundings over the layouts (VPTQ_GEMV_EXACT, the product default arithmetic; round 5)
EXACT_CASES = [
    (1024, 256, dict(), 0),                                            # 8 slices
    (2048, 1032, dict(bias=True), 3),
    (1024, 4608, dict(dist="llm", bias=True), 18),                     # more than 16 rows per wave
    (4104, 264, dict(dist="llm"), 0),                                  # 8 slices without, 8 with the residual table (4704)
    (4712, 136, dict(dist="llm"), 0),                                  # 8 slices without the residual table, 16 with it
    (64, 72, dict(), 0),
    (8192, 512, dict(dist="llm", bias=True), 0),                       # 16 slices (scale, bias and x of 8192 columns: 48 KiB)
    (2048, 520, dict(dist="llm", enable_perm=True, bias=True), 0),     # a permutation: x gathered, scale / bias in column order
    (6152, 264, dict(enable_perm=True), 2),
    (14336, 128, dict(dist="llm"), 0),                                 # 16 slices, two staging rounds
    (16392, 72, dict(dist="llm"), 0),                                  # too wide for the reference's roundings in one piece: 3 column parts
    (28672, 136, dict(dist="llm", bias=True), 0),                      # 2 column parts of 14336 (the down projection of the 70B class)
    (24576, 72, dict(dist="llm", enable_perm=True), 2),                # 2 x 12288, each part its slice of the permutation
    (32768, 72, dict(dist="llm"), 0),                                  # 2 x 16384: a part is still too wide; 3 do not divide it: not served
]


@pytest.mark.parametrize("v,kr", [(8, 0), (8, 256), (16, 0), (8, 65536), (16, 65536), (8, 4096), (16, 1024), (8, 4)])
@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("I,O,kw,rpw", EXACT_CASES)
def test_sliced_layout_reference_roundings(I, O, kw, rpw, dt, v, kr, dev):
    """w = f16(f16(f16(c + r) * s) + b) per weight, as the reference CPU path rounds it, with LDS-local gathers: against the
    oracle (almost every output bit-identical, the rest one flip of the last bit) and against the gather kernel, which
    evaluates the same form through the caches"""
    from vptq_amd.utils.sliced import SlicedGemv
    from vptq_amd import _backend as B
    kw = dict(kw)
    dist = kw.pop("dist", "ref-test")
    L = vo.make_layer(I, O, dist=dist, seed=I + O + v, dtype=dt, vector_len=v, num_centroids=65536, num_res_centroids=kr, **kw)
    x = _x(I, dt, dist, I + 1)
    m = spec_to_module(L, dev)
    desc = m._descriptor()[1]
    want_slices = B.lib().vptq_sliced_layout_supported_for(desc, EXACT)
    small = 8 if v == 8 else 16
    lds = lambda nsl: (65536 // nsl) * v * 2 + (I + 64) * 6 + 64 + (4096 if kr == 256 else 0)   # noqa: E731
    assert want_slices == (small if lds(small) <= 163840 else (2 * small if lds(2 * small) <= 163840 else 0))
    from vptq_amd.utils.sliced import exact_column_parts
    parts, pslices = exact_column_parts(desc, I)
    if not parts:
        with pytest.raises(ValueError):
            SlicedGemv(m, exact=True)
        return
    sl = SlicedGemv(m, rows_per_wave=rpw, exact=True)
    # (too wide for 6 bytes of LDS per column in one piece: equal column parts of a multiple of 8 columns - 16392 = 3 x 5464)
    assert sl.exact and sl.parts == parts == (1 if want_slices else 3 if I == 16392 else 2) and sl.slices == (want_slices or pslices)
    xt = bits_to_tensor(x, dt, dev).reshape(x.shape)
    got = sl(xt)
    torch.cuda.synchronize()
    want = vo.forward(L, x)
    gb = tensor_to_bits(got)
    assert rel_err(gb, want, dt) <= TOL[dt], f"{I}x{O} {dt}: {rel_err(gb, want, dt):.3e}"
    ident = float((gb.reshape(-1) == np.asarray(want).reshape(-1)).mean())
    assert ident >= 0.95, ident
    ref = tensor_to_bits(gemv_abi(m, xt, EXACT))                      # gather kernel: the same weights, another summation order
    assert float((gb.reshape(-1) == ref.reshape(-1)).mean()) >= 0.95
    # fp32 outputs: one rounding of the same sums; determinism; several tokens are not this object's business
    y32 = sl(xt, flags=B.GEMV_OUT_F32)
    assert y32.dtype == torch.float32 and torch.equal(y32.to(got.dtype).view(torch.int16), got.view(torch.int16))
    assert torch.equal(sl(xt).view(torch.int16), got.view(torch.int16))
    # two tables of v = 16, two tables of v = 8 whose operands of two tokens do not fit beside the slice: one token only; column parts:
    # 2 / 3 tokens where every part takes them in one pass (whole or in window parts), nothing else
    if (v == 16 and kr) or (kr not in (0, 256) and not sl.tokens_one_pass(2)) or (sl.parts > 1 and not sl.tokens_one_pass(2)):
        assert not sl.tokens_supported(2) and sl.forward_tokens(torch.cat([xt, xt], dim=1)) is None
    elif sl.parts > 1:
        assert sl.tokens_supported(2) and not sl.tokens_supported(4)
        x2 = _xt(I, 2, dt, dist, I + 9)
        got2 = sl.forward_tokens(bits_to_tensor(x2, dt, dev).reshape(x2.shape))
        torch.cuda.synchronize()
        want2 = vo.forward(L, x2)
        g2 = tensor_to_bits(got2)
        assert got2.shape == (1, 2, O) and rel_err(g2, want2, dt) <= TOL[dt], f"{I}x{O} {dt} 2 tokens over {sl.parts} column parts: {rel_err(g2, want2, dt):.3e}"
        assert float((g2.reshape(-1) == np.asarray(want2).reshape(-1)).mean()) >= 0.95
        assert sl.forward_tokens(torch.cat([xt] * 4, dim=1)) is None


@pytest.mark.parametrize("scale", [1e-3, 1e-4])
@pytest.mark.parametrize("v,kr", [(8, 0), (8, 256), (8, 65536)])
@pytest.mark.parametrize("dt", ["f16", "bf16"])
def test_sliced_reference_roundings_small_outputs(dt, v, kr, scale, dev):
    """the slices' partial sums meet in fixed-point accumulator words.  Round 5's unit was 2^-24 with floor - an absolute
    error of up to arrivals x 6e-8, always downwards, whatever |y| is: outputs of magnitude 1e-3 and below lost the >= 95 %
    bit-identity of the exact route (and a float32 output its low bits).  Activations scaled by 1e-3 / 1e-4 (outputs ~1e-2 ... 1e-4):
    the same bars as at magnitude 1."""
    from vptq_amd.utils.sliced import SlicedGemv
    from vptq_amd import _backend as B
    I, O = 8192, 520
    L = vo.make_layer(I, O, dist="llm", seed=77 + v + kr, dtype=dt, vector_len=v, num_centroids=65536, num_res_centroids=kr)
    x = vo.from_f32((vo.to_f32(_x(I, dt, "llm", 5), dt) * scale).astype(np.float32), dt)
    m = spec_to_module(L, dev)
    sl = SlicedGemv(m, exact=True)
    xt = bits_to_tensor(x, dt, dev).reshape(x.shape)
    got = sl(xt)
    torch.cuda.synchronize()
    want = vo.forward(L, x)
    gb = tensor_to_bits(got)
    assert float(np.abs(vo.to_f32(np.asarray(want), dt)).max()) < 60 * scale      # (the outputs ARE small)
    assert rel_err(gb, want, dt) <= TOL[dt]
    ident = float((gb.reshape(-1) == np.asarray(want).reshape(-1)).mean())
    assert ident >= 0.95, f"{dt} v{v} kr{kr} x{scale:g}: {ident:.4f} bit-identical"
    ref = tensor_to_bits(gemv_abi(m, xt, EXACT))                      # gather kernel: float partial sums, no fixed point
    assert float((gb.reshape(-1) == ref.reshape(-1)).mean()) >= 0.95
    # float32 outputs: against the gather kernel's un-rounded sums, relative to the outputs' OWN magnitude
    y32 = sl(xt, flags=B.GEMV_OUT_F32).reshape(-1).double()
    r32 = gemv_abi(m, xt, EXACT, out_f32=True).reshape(-1).double()
    # (the words' resolution: 2^-30 per arrival for fp16 layers - up to 32 arrivals here -, 2^-28 for bf16 layers)
    res = 32 * (2.0 ** -30 if dt == "f16" else 2.0 ** -28)
    assert float((y32 - r32).abs().max()) <= max(2e-6 * float(r32.abs().max()), res)


EXACT_TOK_SHAPES = [(2048, 528, dict(dist="llm", enable_perm=True, bias=True)), (8192, 512, dict(dist="llm", bias=True)), (4104, 272, dict(dist="llm")),
                    (14336, 136, dict(dist="llm")), (72, 1040, dict()), (4096, 2056, dict(dist="llm", enable_perm=True))]


@pytest.mark.parametrize("tokens,dt", [(2, "f16"), (3, "bf16"), (4, "f16"), (4, "bf16"), (5, "f16"), (7, "bf16"), (8, "f16"), (8, "bf16")])
@pytest.mark.parametrize("v,k,kr", [(8, 65536, 0), (8, 65536, 256), (16, 65536, 0), (8, 16384, 256), (8, 65536, 65536), (8, 65536, 1024)])
@pytest.mark.parametrize("I,O,kw", EXACT_TOK_SHAPES)
def test_sliced_tokens_reference_roundings(I, O, kw, v, k, kr, tokens, dt, dev):
    """2 - 8 tokens in ONE launch over the exact sliced layout, every weight rebuilt as f16(f16(f16(c + r) s) + b) in the matrix
    pipe's operand layout: against the oracle (almost every output bit-identical), against the gather kernel in the same
    arithmetic, against the one-token exact sliced kernel; float32 outputs; repeatable"""
    from vptq_amd.utils.sliced import SlicedGemv
    from vptq_amd import _backend as B
    kw = dict(kw)
    dist = kw.pop("dist", "ref-test")
    L = vo.make_layer(I, O, dist=dist, seed=I + O + v + kr + tokens, dtype=dt, vector_len=v, num_centroids=k, num_res_centroids=kr, **kw)
    x = _xt(I, tokens, dt, dist, I + 7)
    m = spec_to_module(L, dev)
    sl = SlicedGemv(m, exact=True)
    xt = bits_to_tensor(x, dt, dev).reshape(x.shape)
    if kr not in (0, 256) and not sl.tokens_one_pass(tokens):   # two tables: 2 / 3 tokens in one pass (residual entries gathered once), else none
        assert not sl.tokens_supported(tokens) and sl.forward_tokens(xt) is None
        # (what one pass needs: the slice + (2 tokens + 4) bytes per column within 160 KiB, at most 3 tokens)
        assert tokens > 3 or (k // sl.slices) * v * 2 + (I + 64) * (4 + 2 * tokens) + 64 > 163840, (I, tokens)
        return
    assert sl.tokens_supported(tokens)    # (8 + 2 x token slots bytes per column and phase: every width of these cases fits)
    got = sl.forward_tokens(xt)
    torch.cuda.synchronize()
    assert got.shape == (1, tokens, O)
    want = vo.forward(L, x)
    gb = tensor_to_bits(got)
    err = rel_err(gb, want, dt)
    assert err <= TOL[dt], f"v{v}-k{k}-{kr} {I}x{O} {tokens} tokens {dt}: {err:.3e}"
    assert float((gb.reshape(-1) == np.asarray(want).reshape(-1)).mean()) >= 0.95
    ref = tensor_to_bits(gemv_abi(m, xt, EXACT))
    assert float((gb.reshape(-1) == ref.reshape(-1)).mean()) >= 0.95
    for t in (0, tokens - 1):
        one = sl(xt[:, t:t + 1].contiguous())
        assert float((tensor_to_bits(got[:, t:t + 1]).reshape(-1) == tensor_to_bits(one).reshape(-1)).mean()) >= 0.95
    y32 = sl.forward_tokens(xt, flags=B.GEMV_OUT_F32)
    assert y32.dtype == torch.float32 and torch.equal(y32.to(got.dtype).view(torch.int16), got.view(torch.int16))
    assert torch.equal(sl.forward_tokens(xt).view(torch.int16), got.view(torch.int16))
    # 2 / 3 tokens where the slice + (2 tokens + 4) bytes per column fit the LDS: ONE pass of the one-token kernel (TOK) - it needs
    # no column windows; everything else walks them in phases
    lib = B.lib()
    lay = B.SlicedLayout.from_buffer_copy(sl.layout[0])
    lay.wstart = None
    tab = (k // sl.slices) * v * 2
    one_pass = tokens <= (3 if v == 8 else 2) and tab + (I + 64) * (4 + 2 * tokens) + 64 + (4096 if kr == 256 else 0) <= 163840
    assert bool(lib.vptq_quant_gemv_sliced_tokens_supported_for(sl.desc, lay, tokens, EXACT)) == one_pass
    if one_pass:
        need = lib.vptq_quant_gemv_sliced_tokens_workspace_bytes(sl.desc, tokens)
        ws = torch.zeros(need, dtype=torch.uint8, device=dev)
        y = torch.empty_like(got)
        assert lib.vptq_quant_gemv_sliced_tokens(sl.desc, lay, xt.data_ptr(), y.data_ptr(), tokens, EXACT, ws.data_ptr(), need, B.current_stream_ptr(dev)) == 0
        torch.cuda.synchronize()
        assert torch.equal(y.view(torch.int16), got.view(torch.int16))
        nrows = (O + v - 1) // v
        cnt = ((nrows + 15) // 16 * 4 + 255) // 256 * 256
        assert int(ws[cnt:cnt + 3 * nrows * v * 8].count_nonzero().item()) == 0


@pytest.mark.parametrize("name,parts", [("t8_k65536_r256", [(0, 4), (0, 8), (2, 4), (3, 8)]), ("t2_k65536_r256_4096x4096", [(0, 2)]),
                                        ("t4_k65536_r0_8192x2048_perm", [(0, 4), (1, 4), (2, 4)])])
def test_sliced_tokens_reference_roundings_on_reference_goldens(name, parts, dev):
    """the real reference's outputs for 2 - 8 tokens of one-table k = 65536 layers"""
    from vptq_amd.utils.sliced import SlicedGemv
    L, x, y, cfg, _ = load_fmt(name)
    dt = cfg["dtype"]
    sl = SlicedGemv(spec_to_module(L, dev), exact=True)
    xt = bits_to_tensor(x, dt, dev).reshape(x.shape)
    for a, b in parts:
        got = sl.forward_tokens(xt[:, a:b].contiguous())
        assert got is not None, (name, a, b)
        gb = tensor_to_bits(got)
        assert rel_err(gb, y[:, a:b], dt) <= TOL[dt], (name, a, b)
        assert float((gb.reshape(-1) == np.asarray(y[:, a:b]).reshape(-1)).mean()) >= 0.95, (name, a, b)


@pytest.mark.parametrize("v,kr", [(8, 2), (8, 64), (16, 4), (8, 0), (8, 256)])
def test_sliced_layouts_with_every_element_in_one_slice(v, kr, dev):
    """all main indices in ONE slice (the others' lists are empty: their waves have no stream at all) and a tiny residual
    table: a wave without blocks must not read behind the last list - the residual index it would pick up there sends the
    exact kernel's L2 gather anywhere (a memory fault found by tools/gpu_fuzz.py --sliced in round 5)"""
    from vptq_amd.utils.sliced import SlicedGemv
    I, O = 3384, 4684 // v * v
    L = vo.make_layer(I, O, dist="llm", seed=91 + kr, vector_len=v, num_centroids=65536, num_res_centroids=kr)
    N = L.indices.shape[1]
    rng = np.random.default_rng(kr)
    idx = (rng.integers(0, 8192, size=(1, N, I), dtype=np.int64)) | (3 << 13)
    ridx = None if kr == 0 else np.full((1, N, I), kr - 1, dtype=np.int64)
    L.indices = vo.pack_indices(idx, L.index_bits, ridx, L.res_bits)
    m = spec_to_module(L, dev)
    x = _x(I, "f16", "llm", 17)
    xt = bits_to_tensor(x, "f16", dev).reshape(x.shape)
    want = vo.forward(L, x)
    for exact in (False, True):
        for rpw in (0, 5):
            sl = SlicedGemv(m, rows_per_wave=rpw, exact=exact)
            got = sl(xt)
            torch.cuda.synchronize()
            assert rel_err(tensor_to_bits(got), want, "f16") <= 1e-3, (exact, rpw)


def test_module_default_route_takes_the_exact_sliced_kernel(dev):
    """the product default (reference roundings): one-token calls of v8-k65536-0 / -256 layers go over an EXACT sliced layout,
    two-table formats and several tokens keep the gather kernel; siblings share one launch"""
    import vptq_amd
    from vptq_amd.layers.vqlinear import SiblingGroup
    assert vptq_amd.arithmetic() == "reference"
    L = vo.make_layer(2048, 512, seed=31, dist="llm", num_centroids=65536, num_res_centroids=256, bias=True)
    m = spec_to_module(L, dev)
    x1 = _x(2048, "f16", "llm", 3)
    xt = bits_to_tensor(x1, "f16", dev).reshape(x1.shape)
    y0 = m(xt)                                # a layer this small stays on the gather kernel by itself (under 1 M index elements)
    assert m.__dict__["_sliced"][1] is None and torch.equal(y0.view(torch.int16), gemv_abi(m, xt, EXACT).view(torch.int16))
    m.enable_sliced_layout()
    y = m(xt)
    sl = m.__dict__["_sliced"][1]
    assert sl is not None and sl.exact and sl.slices == 8
    want = vo.forward(L, x1)
    assert rel_err(tensor_to_bits(y), want, "f16") <= 1e-3
    assert float((tensor_to_bits(y).reshape(-1) == np.asarray(want).reshape(-1)).mean()) >= 0.95
    x3 = torch.cat([xt, xt, xt], dim=1)
    y3 = m(x3)                                                          # three tokens: the gather kernel (same arithmetic)
    assert torch.equal(y3.view(torch.int16), gemv_abi(m, x3, EXACT).view(torch.int16))
    # a two-table format: ONE exact layout, the residual entries gathered from device memory
    L2 = vo.make_layer(2048, 512, seed=32, dist="llm", num_centroids=65536, num_res_centroids=65536)
    m2 = spec_to_module(L2, dev)
    m2.enable_sliced_layout()
    y2 = m2(xt)
    s2 = m2.__dict__["_sliced"][1]
    assert s2 is not None and s2.exact and len(s2.layout) == 1 and s2.res.dtype == torch.int16
    assert rel_err(tensor_to_bits(y2), vo.forward(L2, x1), "f16") <= 1e-3
    assert float((y2.view(torch.int16) == gemv_abi(m2, xt, EXACT).view(torch.int16)).float().mean()) >= 0.95
    # siblings (q / k / v): one launch of the exact kernel, the members' own bits
    Ls = [vo.make_layer(2048, O, seed=40 + i, dist="llm", num_centroids=65536, num_res_centroids=256, bias=(i == 1)) for i, O in enumerate((1024, 264, 512))]
    ms = [spec_to_module(Li, dev) for Li in Ls]
    for mm in ms:
        mm.enable_sliced_layout()
    alone = [mm(xt) for mm in ms]
    group = SiblingGroup(ms)
    for mm in ms:
        object.__setattr__(mm, "_siblings", group)
    ys = [mm(xt) for mm in ms]
    assert isinstance(group.__dict__.get("_sgroup"), tuple) and group._sgroup[1].exact
    for a, b, Li in zip(alone, ys, Ls):
        assert torch.equal(a.view(torch.int16), b.view(torch.int16))
        assert rel_err(tensor_to_bits(b), vo.forward(Li, x1), "f16") <= 1e-3


@pytest.mark.parametrize("name", [n for n in fmt_names() if ("k65536_r0" in n or "k65536_r256" in n) and n.startswith("t1_") and "v16" not in n])
def test_sliced_layout_reference_roundings_on_reference_goldens(name, dev):
    """one-token goldens of the real reference through the exact sliced kernel: the reference's own outputs, bit for bit
    on almost every element"""
    from vptq_amd.utils.sliced import SlicedGemv
    L, x, y, cfg, _ = load_fmt(name)
    dt = cfg["dtype"]
    m = spec_to_module(L, dev)
    xt = bits_to_tensor(x, dt, dev).reshape(x.shape)
    sl = SlicedGemv(m, exact=True)
    got = tensor_to_bits(sl(xt[:, :1].contiguous()))
    assert rel_err(got, y[:, :1], dt) <= TOL[dt]
    assert float((got.reshape(-1) == np.asarray(y[:, :1]).reshape(-1)).mean()) >= 0.95


@pytest.mark.parametrize("name", [n for n in fmt_names() if "k65536_r0" in n or "k65536_r256" in n or "k65536_r65536" in n])
def test_sliced_layout_gemv_on_reference_goldens(name, dev):
    """v8-k65536-0 / -256 / -65536 layers whose y comes from the real reference (tests/golden/gen_golden_fmt.py)"""
    from vptq_amd.utils.sliced import SlicedGemv
    L, x, y, cfg, _ = load_fmt(name)
    dt = cfg["dtype"]
    m = spec_to_module(L, dev)
    xt = bits_to_tensor(x, dt, dev).reshape(x.shape)
    sl = SlicedGemv(m)
    for t in range(min(cfg["tokens"], 3)):   # (a fixture of several tokens: one at a time)
        got = tensor_to_bits(sl(xt[:, t:t + 1].contiguous()))
        err = rel_err(got, y[:, t:t + 1], dt)
        assert err <= TOL[dt], f"{name} token {t}: {err:.3e}"
    # ... and the library's default route on the same fixture
    assert rel_err(tensor_to_bits(gemv_abi(m, xt, 0)), y, dt) <= TOL[dt]


def test_sliced_layout_small_reference_golden_and_rejections(dev):
    from vptq_amd.utils.sliced import SlicedGemv
    from vptq_amd import _backend as B
    L, x, y, cfg, _ = load_golden("k65536_nores_nonorm")
    m = spec_to_module(L, dev)
    if B.lib().vptq_sliced_layout_supported(m._descriptor()[1]):
        xt = bits_to_tensor(x, cfg["dtype"], dev).reshape(x.shape)
        got = tensor_to_bits(SlicedGemv(m)(xt[:, :1].contiguous()))
        assert rel_err(got, y[:, :1], cfg["dtype"]) <= TOL[cfg["dtype"]]
    else:   # (no weight_scale / weight_bias: not this path's layer)
        with pytest.raises(ValueError):
            SlicedGemv(m)
    # a canonical 256 + 256 layer is not a sliced-layout layer
    with pytest.raises(ValueError):
        SlicedGemv(spec_to_module(vo.make_layer(256, 64, seed=1), dev))
    # the reference's roundings (ABI 8) need a layout built for THAT arithmetic: the folded form's pair of layouts of a
    # two-table layer carries no 16-bit residual stream, an 8-slice folded layout of an 8192-column layer has the wrong slice
    # count ("unsupported" from the library = None here: the caller takes the regular route, which has the roundings)
    L2 = vo.make_layer(512, 128, seed=2, num_centroids=65536, num_res_centroids=65536)
    sl2 = SlicedGemv(spec_to_module(L2, dev))
    assert sl2(torch.zeros(1, 1, 512, dtype=torch.float16, device=dev), flags=EXACT) is None
    Lw = vo.make_layer(8192, 64, seed=3, num_centroids=65536, num_res_centroids=0)
    mw = spec_to_module(Lw, dev)
    slw = SlicedGemv(mw)
    assert slw.slices == 8 and B.lib().vptq_sliced_layout_supported_for(mw._descriptor()[1], EXACT) == 16
    assert slw(torch.zeros(1, 1, 8192, dtype=torch.float16, device=dev), flags=EXACT) is None


def test_sliced_layout_in_a_hipgraph(dev):
    from vptq_amd.utils.sliced import SlicedGemv
    L = vo.make_layer(2048, 1024, seed=9, dist="llm", num_centroids=65536, num_res_centroids=0)
    m = spec_to_module(L, dev)
    sl = SlicedGemv(m)
    xs = torch.zeros(1, 1, 2048, dtype=torch.float16, device=dev)
    ys = torch.empty(1, 1, 1024, dtype=torch.float16, device=dev)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        sl(xs, ys)          # warm-up on the capture stream: the layer's workspace is per stream
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            assert sl(xs, ys) is ys
    for rep in range(2):
        v = _x(2048, "f16", "llm", 50 + rep)
        xs.copy_(bits_to_tensor(v, "f16", dev).reshape(1, 1, 2048))
        g.replay()
        torch.cuda.synchronize()
        assert rel_err(tensor_to_bits(ys), vo.forward(L, v), "f16") <= 1e-3


def test_module_forward_takes_the_sliced_layout_when_enabled(dev, folded_arithmetic):
    """VQuantLinear.enable_sliced_layout() - the default (VPTQ_SLICED_LAYOUT=auto) while the layout leaves a quarter
    of the device memory free: one-token calls go through the derived layout, everything else (several tokens,
    dequant) through the state-dict tensors as before; an in-place change of the indices rebuilds it"""
    import vptq_amd.layers.vqlinear as vq
    L0 = vo.make_layer(1024, 256, seed=20, dist="llm", num_centroids=65536, num_res_centroids=0)
    m0 = spec_to_module(L0, dev)
    x0 = _x(1024, "f16", "llm", 4)
    y0 = m0(bits_to_tensor(x0, "f16", dev).reshape(x0.shape))          # nothing enabled by hand: auto
    if vq._SLICED_LAYOUT_ENV:
        assert m0.__dict__["_sliced"][1] is not None
    assert rel_err(tensor_to_bits(y0), vo.forward(L0, x0), "f16") <= 1e-3
    L = vo.make_layer(2048, 512, seed=21, dist="llm", num_centroids=65536, num_res_centroids=256, bias=True)
    m = spec_to_module(L, dev)
    x1 = _x(2048, "f16", "llm", 3)
    xt = bits_to_tensor(x1, "f16", dev).reshape(x1.shape)
    m.enable_sliced_layout(False)
    y_default = m(xt)
    assert m.__dict__.get("_sliced") is None and m._sliced_gemv() is None
    assert torch.equal(y_default.view(torch.int16), gemv_abi(m, xt, 0).view(torch.int16))   # the gather kernel
    m.enable_sliced_layout()
    y_sliced = m(xt)
    assert m.__dict__["_sliced"][1] is not None
    assert rel_err(tensor_to_bits(y_sliced), vo.forward(L, x1), "f16") <= 1e-3
    assert rel_err(tensor_to_bits(y_sliced), tensor_to_bits(y_default), "f16") <= 1e-3
    x3 = torch.cat([xt, xt, xt], dim=1)
    y3 = m(x3)   # three tokens: the gather kernel
    assert torch.equal(y3[:, :1].view(torch.int16), y_default.view(torch.int16))
    # new indices (another storage): the layout follows
    L2 = vo.make_layer(2048, 512, seed=22, dist="llm", num_centroids=65536, num_res_centroids=256, bias=True)
    m.indices.data = torch.from_numpy(L2.indices.copy()).to(dev).reshape(m.indices.shape)
    L.indices = L2.indices
    assert rel_err(tensor_to_bits(m(xt)), vo.forward(L, x1), "f16") <= 1e-3
    # ... also after an in-place rewrite of the same storage
    L3 = vo.make_layer(2048, 512, seed=23, dist="llm", num_centroids=65536, num_res_centroids=256, bias=True)
    with torch.no_grad():
        m.indices.copy_(torch.from_numpy(L3.indices.copy()).to(dev).reshape(m.indices.shape))
    L.indices = L3.indices
    assert rel_err(tensor_to_bits(m(xt)), vo.forward(L, x1), "f16") <= 1e-3
    m.enable_sliced_layout(False)
    assert m._sliced_gemv() is None


def test_sliced_route_falls_back_instead_of_failing(dev, folded_arithmetic):
    """the sliced one-token route is on by default and sits in front of the regular routing, so whatever it
    cannot take must fall through to it: an activation that is not 16-byte aligned, a capture on a stream the layer
    has never run on (its workspace is per stream and is not allocated inside a capture), a first call inside a
    capture (no layout is built there - and that "no" must not stick)."""
    L = vo.make_layer(2048, 512, seed=31, dist="llm", num_centroids=65536, num_res_centroids=256)
    m = spec_to_module(L, dev)
    m.enable_sliced_layout()
    x1 = _x(2048, "f16", "llm", 5)
    want = vo.forward(L, x1)
    buf = torch.zeros(2048 + 8, dtype=torch.float16, device=dev)
    xm = buf[1:2049].view(1, 1, 2048)                 # contiguous, 2 bytes off a 16-byte boundary
    xm.copy_(bits_to_tensor(x1, "f16", dev).reshape(1, 1, 2048))
    assert xm.data_ptr() % 16 == 2
    # a first ONE-TOKEN call inside a capture (the descriptor exists: a prompt went through the layer before - building it
    # reads the load-time gate back from the device, which no capture allows): regular route, nothing remembered
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    xa = bits_to_tensor(x1, "f16", dev).reshape(1, 1, 2048)
    m(torch.cat([xa] * 5, dim=1))          # (5 tokens: the gather kernel; 1 - 4 tokens would build the layout here)
    torch.cuda.synchronize()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            yg = m(xa)
    assert m.__dict__.get("_sliced") is None
    g.replay()
    torch.cuda.synchronize()
    assert rel_err(tensor_to_bits(yg), want, "f16") <= 1e-3
    ya = m(xa)                                        # builds the layout now
    sl = m.__dict__["_sliced"][1]
    assert sl is not None
    assert rel_err(tensor_to_bits(ya), want, "f16") <= 1e-3
    ym = m(xm)                                        # misaligned: the gather kernel
    assert rel_err(tensor_to_bits(ym), want, "f16") <= 1e-3
    assert sl(xm) is None
    # capture on a stream this layer has not run on: no workspace is created inside the capture
    g2 = torch.cuda.CUDAGraph()
    s2 = torch.cuda.Stream()
    with torch.cuda.stream(s2):
        with torch.cuda.graph(g2, stream=s2):
            assert sl(xa) is None
            y2 = m(xa)
    g2.replay()
    torch.cuda.synchronize()
    assert rel_err(tensor_to_bits(y2), want, "f16") <= 1e-3
    # two streams: separate workspaces
    with torch.cuda.stream(s2):
        y3 = m(xa)
    torch.cuda.synchronize()
    assert len(sl._ws) >= 2 and torch.equal(y3.view(torch.int16), ya.view(torch.int16))


V16_CASES = [
    (1024, 256, dict(), 0),
    (2048, 1040, dict(bias=True), 3),
    (4104, 272, dict(dist="llm"), 0),
    (64, 72, dict(), 0),                                     # O not a multiple of 16: a padded last vector-row
    (8192, 512, dict(dist="llm", bias=True), 1),
    (2048, 528, dict(dist="llm", enable_perm=True, bias=True), 0),
    (14352, 128, dict(dist="llm"), 0),                       # wider than 14336 columns: 32 slices of 2048 entries
]


@pytest.mark.parametrize("kr", [0, 65536])
@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("I,O,kw,rpw", V16_CASES)
def test_sliced_layout_gemv_vector_length_16(I, O, kw, rpw, dt, kr, dev):
    """v16-k65536-0 and v16-k65536-65536 (the "2 bits" format of most published model families): 32-byte entries, 16 / 32
    slices; the two-table format as two passes, one per table"""
    from vptq_amd.utils.sliced import SlicedGemv
    from vptq_amd import _backend as B
    kw = dict(kw)
    dist = kw.pop("dist", "ref-test")
    L = vo.make_layer(I, O, dist=dist, seed=I + O + 16, dtype=dt, vector_len=16, num_centroids=65536, num_res_centroids=kr, **kw)
    x = _x(I, dt, dist, I + 1)
    m = spec_to_module(L, dev)
    assert kernel_name(m, 1) == "gemv_gatherx_kernel"
    sl = SlicedGemv(m, rows_per_wave=rpw)
    assert sl.slices == (16 if I <= 14336 else 32) and len(sl.layout) == (2 if kr else 1)
    xt = bits_to_tensor(x, dt, dev).reshape(x.shape)
    got = sl(xt)
    torch.cuda.synchronize()
    want = vo.forward(L, x)
    err = rel_err(tensor_to_bits(got), want, dt)
    assert err <= TOL[dt], f"{I}x{O} {dt}: {err:.3e}"
    assert rel_err(tensor_to_bits(got), tensor_to_bits(gemv_abi(m, xt, 0)), dt) <= TOL[dt]
    y32 = sl(xt, flags=B.GEMV_OUT_F32)
    assert y32.dtype == torch.float32 and torch.equal(y32.to(got.dtype).view(torch.int16), got.view(torch.int16))
    assert torch.equal(sl(xt).view(torch.int16), got.view(torch.int16))
    # the module's own one-token route takes it
    m.enable_sliced_layout()
    assert torch.equal(m(xt).view(torch.int16), got.view(torch.int16)) or rel_err(tensor_to_bits(m(xt)), want, dt) <= TOL[dt]


@pytest.mark.parametrize("name", [n for n in fmt_names() if "v16_k65536" in n and n.startswith("t1_")])
def test_sliced_layout_vector_length_16_on_reference_goldens(name, dev):
    from vptq_amd.utils.sliced import SlicedGemv
    L, x, y, cfg, _ = load_fmt(name)
    dt = cfg["dtype"]
    m = spec_to_module(L, dev)
    xt = bits_to_tensor(x, dt, dev).reshape(x.shape)
    got = tensor_to_bits(SlicedGemv(m)(xt))
    assert rel_err(got, y, dt) <= TOL[dt], name


# (vector length, main centroids, residual centroids): the other members of the family among the published checkpoints -
# a residual codebook of any size is a second table in the same launch (small ones held whole by their workgroups), main
# codebooks of 16384 / 32768 entries give smaller slices
FAMILY = [(8, 65536, 4), (8, 65536, 1024), (8, 65536, 4096), (8, 32768, 0), (8, 16384, 0), (16, 65536, 64), (16, 65536, 256),
          (16, 65536, 1024), (16, 65536, 16384), (16, 32768, 32768)]
FAMILY_SHAPES = [(2048, 528, dict(dist="llm", enable_perm=True, bias=True)), (4104, 272, dict(dist="llm")), (72, 1040, dict()),
                 (8192, 512, dict(dist="llm", bias=True))]


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("v,k,kr", FAMILY)
@pytest.mark.parametrize("I,O,kw", FAMILY_SHAPES)
def test_sliced_layout_family(I, O, kw, v, k, kr, dt, dev):
    from vptq_amd.utils.sliced import SlicedGemv
    from vptq_amd import _backend as B
    kw = dict(kw)
    dist = kw.pop("dist", "ref-test")
    L = vo.make_layer(I, O, dist=dist, seed=I + O + v + kr, dtype=dt, vector_len=v, num_centroids=k, num_res_centroids=kr, **kw)
    x = _x(I, dt, dist, I + 2)
    m = spec_to_module(L, dev)
    sl = SlicedGemv(m)
    assert len(sl.layout) == (2 if kr else 1)
    xt = bits_to_tensor(x, dt, dev).reshape(x.shape)
    got = sl(xt)
    torch.cuda.synchronize()
    want = vo.forward(L, x)
    err = rel_err(tensor_to_bits(got), want, dt)
    assert err <= TOL[dt], f"v{v}-k{k}-{kr} {I}x{O} {dt}: {err:.3e}"
    m.enable_sliced_layout(False)
    assert rel_err(tensor_to_bits(got), tensor_to_bits(m(xt)), dt) <= TOL[dt]     # the gather kernels
    y32 = sl(xt, flags=B.GEMV_OUT_F32)
    assert torch.equal(y32.to(got.dtype).view(torch.int16), got.view(torch.int16))
    assert torch.equal(sl(xt).view(torch.int16), got.view(torch.int16))


@pytest.mark.parametrize("name", ["t1_v16_k65536_r1024_4096x4096", "t1_v8_k65536_r4_bias", "t1_v8_k32768_r0_perm"])
def test_sliced_layout_family_on_reference_goldens(name, dev):
    from vptq_amd.utils.sliced import SlicedGemv
    L, x, y, cfg, _ = load_fmt(name)
    dt = cfg["dtype"]
    m = spec_to_module(L, dev)
    xt = bits_to_tensor(x, dt, dev).reshape(x.shape)
    assert rel_err(tensor_to_bits(SlicedGemv(m)(xt)), y, dt) <= TOL[dt], name
    assert rel_err(tensor_to_bits(m(xt)), y, dt) <= TOL[dt], name      # the module's own one-token route (auto: sliced)


def test_two_and_three_tokens_as_sliced_launches_per_token(dev, monkeypatch, folded_arithmetic):
    """2 tokens of a LARGE large-codebook layer (the 4-bit format: 3) are served as one sliced launch per token
    (VQuantLinear._sliced_token_limit: the gather kernels cost as much for one token as for four, the sliced kernel a
    third to a half of that per token); smaller layers and more tokens keep the gather kernels"""
    import vptq_amd.layers.vqlinear as vq
    from oracle import c_oracle as co
    monkeypatch.setattr(vq, "_SLICED_ONE_LAUNCH", "0")    # (the route of this test: one launch PER token)
    L = vo.make_layer(8192, 6200, dist="llm", seed=41, num_centroids=65536, num_res_centroids=0)
    m = spec_to_module(L, dev)
    m.enable_sliced_layout()
    x = _x(8192, "f16", "llm", 7)
    x2 = np.concatenate([x, _x(8192, "f16", "llm", 8)], axis=1)
    xt = bits_to_tensor(x2, "f16", dev).reshape(x2.shape)
    y = m(xt)
    sl = m.__dict__["_sliced"][1]
    assert sl is not None and m._sliced_token_limit(sl) == 2
    for t in range(2):
        assert torch.equal(y[:, t].view(torch.int16), sl(xt[:, t:t + 1].contiguous()).view(torch.int16)[0])
        assert rel_err(tensor_to_bits(y[:, t:t + 1]), co.forward(L, x2[:, t:t + 1], quirk=False), "f16") <= 1e-3
    x3 = torch.cat([xt, xt[:, :1]], dim=1)
    y3 = m(x3)                                   # 3 tokens: the gather kernel
    m.enable_sliced_layout(False)
    assert torch.equal(y3.view(torch.int16), m(x3).view(torch.int16))
    # a small two-table layer: limit 1 by the rule; with the override 3 tokens go through the layouts
    Ls = vo.make_layer(1024, 512, dist="llm", seed=42, num_centroids=65536, num_res_centroids=65536)
    ms = spec_to_module(Ls, dev)
    ms.enable_sliced_layout()
    xs = np.concatenate([_x(1024, "f16", "llm", 9 + i) for i in range(3)], axis=1)
    xst = bits_to_tensor(xs, "f16", dev).reshape(xs.shape)
    ms(xst[:, :1].contiguous())
    sls = ms.__dict__["_sliced"][1]
    assert ms._sliced_token_limit(sls) == 1
    monkeypatch.setattr(vq, "_SLICED_TOKENS_ENV", (3, 3))
    sls.__dict__.pop("_token_limit")
    ys = ms(xst)
    for t in range(3):
        assert torch.equal(ys[:, t].view(torch.int16), sls(xst[:, t:t + 1].contiguous()).view(torch.int16)[0])
    assert rel_err(tensor_to_bits(ys), vo.forward(Ls, xs), "f16") <= 1e-3


@pytest.mark.parametrize("v,kr", [(8, 256), (8, 0), (8, 65536), (16, 65536), (16, 1024)])
def test_sibling_layers_share_one_sliced_launch(v, kr, dev, folded_arithmetic, monkeypatch):
    """q / k / v (gate / up) of a large-codebook model read the same activation: one launch of the sliced kernel for the
    group (`vptq_quant_gemv_sliced_grouped` via `SiblingGroup.forward_sliced`), bit-identical to the layers' own launches
    (a row's sums are formed by one wave in the same order whatever the rows per wave; the cross-slice sum is a fixed tree)"""
    from vptq_amd.layers.vqlinear import SiblingGroup
    from vptq_amd.utils.sliced import SlicedGemv, SlicedGroupGemv
    from vptq_amd import _backend as B
    # (this test is about sharing a launch: the measured gate of the folded form - which may turn a 264-output member down -
    # is switched off; tests/test_hip_parity.py::test_adversarial_families_default_route covers the gate)
    monkeypatch.setattr(B, "FOLDED_MAX_PROBE_DISTANCE", {})
    I = 2048
    outs = (1024, 33 * v, 512)     # (33 vector-rows: the load-time gate serves layers with fewer than 32 by the exact kernels)
    Ls = [vo.make_layer(I, O, dist="llm", seed=70 + i + kr % 7, vector_len=v, num_centroids=65536, num_res_centroids=kr,
                        bias=(i == 1), enable_perm=(i == 2)) for i, O in enumerate(outs)]
    ms = [spec_to_module(L, dev) for L in Ls]
    x = _x(I, "f16", "llm", 11)
    xt = bits_to_tensor(x, "f16", dev).reshape(x.shape)
    for m in ms:
        m.enable_sliced_layout()
    alone = [SlicedGemv(m)(xt) for m in ms]
    group = SiblingGroup(ms)
    for m in ms:
        object.__setattr__(m, "_siblings", group)
    ys = [m(xt) for m in ms]                       # the first call launches all three
    torch.cuda.synchronize()
    assert isinstance(group.__dict__.get("_sgroup"), tuple) and isinstance(group._sgroup[1], SlicedGroupGemv)
    for L, y, a in zip(Ls, ys, alone):
        assert torch.equal(y.view(torch.int16), a.view(torch.int16))
        assert rel_err(tensor_to_bits(y), vo.forward(L, x), "f16") <= 1e-3
    # a new activation object: a new launch; the same object with other contents (version bumped): also
    x2 = bits_to_tensor(_x(I, "f16", "llm", 12), "f16", dev).reshape(x.shape)
    y2 = [m(x2) for m in ms]
    assert rel_err(tensor_to_bits(y2[1]), vo.forward(Ls[1], tensor_to_bits(x2)), "f16") <= 1e-3
    x2.copy_(xt)
    y3 = [m(x2) for m in ms]
    for y, a in zip(y3, alone):
        assert torch.equal(y.view(torch.int16), a.view(torch.int16))
    # in a hipGraph (warm-up on the capture stream)
    g, s = torch.cuda.CUDAGraph(), torch.cuda.Stream()
    with torch.cuda.stream(s):
        [m(xt) for m in ms]
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            yg = [m(xt) for m in ms]
    g.replay()
    torch.cuda.synchronize()
    for y, a in zip(yg, alone):
        assert torch.equal(y.view(torch.int16), a.view(torch.int16))


# ---------------------------------------------------------------- 2 - 4 tokens in one launch (gemv_sliced_tok.hip)
TOK_FORMATS = [(8, 65536, 0), (8, 65536, 256), (8, 65536, 65536), (8, 65536, 1024), (8, 32768, 0), (16, 65536, 0), (16, 65536, 65536),
               (16, 65536, 256)]
# one phase (the tokens' activations fit beside the slice) / 2 - 4 phases / a column count that is no multiple of 32 / 16
# (v = 16: 32) slices / a tiny layer (most windows of most lists empty)
TOK_SHAPES = [(2048, 528, dict(dist="llm", enable_perm=True, bias=True)), (8192, 512, dict(dist="llm", bias=True)), (4104, 272, dict(dist="llm")),
              (14336, 136, dict(dist="llm")), (16392, 72, dict(dist="llm", enable_perm=True)), (72, 1040, dict())]


def _xt(I, T, dt, dist, seed):
    rng = np.random.default_rng(seed)
    xs = (0.02 + 0.5 * rng.standard_normal((1, T, I))) if dist == "ref-test" else rng.standard_normal((1, T, I))
    return vo.from_f32(xs.astype(np.float32), dt)


@pytest.mark.parametrize("tokens,dt", [(2, "f16"), (3, "bf16"), (4, "f16"), (4, "bf16")])
@pytest.mark.parametrize("v,k,kr", TOK_FORMATS)
@pytest.mark.parametrize("I,O,kw", TOK_SHAPES)
def test_sliced_tokens_one_launch(I, O, kw, v, k, kr, tokens, dt, dev):
    """2 - 4 tokens over the sliced layouts in ONE launch (column phases): against the oracle, against one sliced launch per
    token, float32 outputs, repeatable"""
    from vptq_amd.utils.sliced import SlicedGemv
    from vptq_amd import _backend as B
    kw = dict(kw)
    dist = kw.pop("dist", "ref-test")
    L = vo.make_layer(I, O, dist=dist, seed=I + O + v + kr + tokens, dtype=dt, vector_len=v, num_centroids=k, num_res_centroids=kr, **kw)
    x = _xt(I, tokens, dt, dist, I + 3)
    m = spec_to_module(L, dev)
    sl = SlicedGemv(m)
    xt = bits_to_tensor(x, dt, dev).reshape(x.shape)
    if not sl.tokens_supported(tokens):
        assert sl.forward_tokens(xt) is None
        pytest.skip("the activations of this many tokens do not fit beside the slice")
    got = sl.forward_tokens(xt)
    torch.cuda.synchronize()
    assert got.shape == (1, tokens, O)
    want = vo.forward(L, x)
    err = rel_err(tensor_to_bits(got), want, dt)
    assert err <= TOL[dt], f"v{v}-k{k}-{kr} {I}x{O} {tokens} tokens {dt}: {err:.3e}"
    for t in range(tokens):     # the one-token kernel over the same layouts (other summation order: the bar, not bits)
        one = sl(xt[:, t:t + 1].contiguous())
        assert rel_err(tensor_to_bits(got[:, t:t + 1]), tensor_to_bits(one), dt) <= TOL[dt]
    y32 = sl.forward_tokens(xt, flags=B.GEMV_OUT_F32)
    assert y32.dtype == torch.float32 and torch.equal(y32.to(got.dtype).view(torch.int16), got.view(torch.int16))
    assert torch.equal(sl.forward_tokens(xt).view(torch.int16), got.view(torch.int16))


@pytest.mark.parametrize("name,parts", [("t8_k65536_r256", [(0, 4), (4, 8), (2, 4), (5, 8)]), ("t5_k65536_r65536_perm", [(0, 2), (2, 5), (1, 5)]),
                                        ("t2_k65536_r256_4096x4096", [(0, 2)]), ("t4_k65536_r0_8192x2048_perm", [(0, 4), (1, 4), (2, 4)]),
                                        ("t3_k65536_r65536_bf16", [(0, 3), (1, 3)]), ("t2_v16_k65536_r65536_4096x2048", [(0, 2)])])
def test_sliced_tokens_on_reference_goldens(name, parts, dev):
    """the real reference's outputs for 5 / 8 tokens of k = 65536 layers, 2 - 4 tokens at a time"""
    from vptq_amd.utils.sliced import SlicedGemv
    L, x, y, cfg, _ = load_fmt(name)
    dt = cfg["dtype"]
    m = spec_to_module(L, dev)
    sl = SlicedGemv(m)
    xt = bits_to_tensor(x, dt, dev).reshape(x.shape)
    for a, b in parts:
        got = sl.forward_tokens(xt[:, a:b].contiguous())
        assert got is not None
        assert rel_err(tensor_to_bits(got), y[:, a:b], dt) <= TOL[dt], (name, a, b)


def test_sliced_tokens_rejections(dev):
    from vptq_amd.utils.sliced import SlicedGemv
    from vptq_amd import _backend as B
    L = vo.make_layer(2048, 512, dist="llm", seed=51, num_centroids=65536, num_res_centroids=256)
    m = spec_to_module(L, dev)
    sl = SlicedGemv(m)
    x = torch.randn(1, 3, 2048, device=dev, dtype=torch.float16)
    assert sl.tokens_supported(2) and sl.tokens_supported(4) and sl.tokens_supported(8) and not sl.tokens_supported(9) and not sl.tokens_supported(1)
    y = torch.empty(1, 3, 512, device=dev, dtype=torch.float16)
    lib = B.lib()
    need = lib.vptq_quant_gemv_sliced_tokens_workspace_bytes(sl.desc, 3)
    ws = torch.zeros(need, dtype=torch.uint8, device=dev)
    sp = B.current_stream_ptr(dev)
    assert lib.vptq_quant_gemv_sliced_tokens(sl.desc, sl.layout, x.data_ptr(), y.data_ptr(), 3, 0, ws.data_ptr(), need - 1, sp) == B.E_WORKSPACE
    assert lib.vptq_quant_gemv_sliced_tokens(sl.desc, sl.layout, x.data_ptr(), y.data_ptr(), 9, 0, ws.data_ptr(), need, sp) == B.E_UNSUPPORTED
    assert lib.vptq_quant_gemv_sliced_tokens(sl.desc, sl.layout, x.data_ptr(), y.data_ptr(), 3, B.GEMV_FORCE_GENERIC, ws.data_ptr(), need, sp) == B.E_UNSUPPORTED
    assert lib.vptq_quant_gemv_sliced_tokens(sl.desc, sl.layout, x.data_ptr(), y.data_ptr(), 3, 0, ws.data_ptr(), need, sp) == 0
    torch.cuda.synchronize()
    assert torch.equal(y, sl.forward_tokens(x))
    # one-table formats: the layout of the folded form IS the exact one where the slice counts agree (they do at 2048 columns) ...
    assert lib.vptq_sliced_layout_supported_for(sl.desc, EXACT) == sl.slices
    assert lib.vptq_quant_gemv_sliced_tokens_supported_for(sl.desc, sl.layout, 3, EXACT)
    assert lib.vptq_quant_gemv_sliced_tokens(sl.desc, sl.layout, x.data_ptr(), y.data_ptr(), 3, EXACT, ws.data_ptr(), need, sp) == 0
    torch.cuda.synchronize()
    assert torch.equal(y, SlicedGemv(m, exact=True).forward_tokens(x))
    # ... two-table formats have no exact token kernel
    L2 = vo.make_layer(2048, 512, dist="llm", seed=52, num_centroids=65536, num_res_centroids=65536)
    sl2 = SlicedGemv(spec_to_module(L2, dev))
    assert lib.vptq_quant_gemv_sliced_tokens_supported_for(sl2.desc, sl2.layout, 3, 0)
    assert not lib.vptq_quant_gemv_sliced_tokens_supported_for(sl2.desc, sl2.layout, 3, EXACT)
    need2 = lib.vptq_quant_gemv_sliced_tokens_workspace_bytes(sl2.desc, 3)
    ws2 = torch.zeros(need2, dtype=torch.uint8, device=dev)
    assert lib.vptq_quant_gemv_sliced_tokens(sl2.desc, sl2.layout, x.data_ptr(), y.data_ptr(), 3, EXACT, ws2.data_ptr(), need2, sp) == B.E_UNSUPPORTED
    # a layout without the column windows' table serves one token only
    lay = B.SlicedLayout.from_buffer_copy(sl.layout[0])
    lay.wstart = None
    assert not lib.vptq_quant_gemv_sliced_tokens_supported(sl.desc, lay, 3)
    assert sl(x[:, :1].contiguous()) is not None


def test_module_route_for_two_to_four_tokens_in_one_launch(dev, monkeypatch, folded_arithmetic):
    """the module's forward: 2 - 4 tokens of a large-codebook layer = ONE launch over the layouts (VQuantLinear._sliced_one_launch:
    where that was measured faster - not v = 16 with a small residual table, not tiny layers); switched off: the other routes"""
    import vptq_amd.layers.vqlinear as vq
    L = vo.make_layer(4096, 1024, dist="llm", seed=61, num_centroids=65536, num_res_centroids=256, bias=True)
    m = spec_to_module(L, dev)
    m.enable_sliced_layout()
    xs = np.concatenate([_x(4096, "f16", "llm", 20 + i) for i in range(5)], axis=1)
    xt = bits_to_tensor(xs, "f16", dev).reshape(xs.shape)
    m(xt[:, :1].contiguous())
    sl = m.__dict__["_sliced"][1]
    assert sl is not None and all(m._sliced_one_launch(sl, t) for t in (2, 3, 4))
    for T in (2, 3, 4):
        x = xt[:, :T].contiguous()
        y = m(x)
        assert torch.equal(y.view(torch.int16), sl.forward_tokens(x).view(torch.int16))
        assert rel_err(tensor_to_bits(y), vo.forward(L, xs[:, :T]), "f16") <= 1e-3
    y5 = m(xt)                                     # 5 tokens: the gather kernel
    assert rel_err(tensor_to_bits(y5), vo.forward(L, xs), "f16") <= 1e-3
    y2 = m(xt[:, :2].contiguous())
    monkeypatch.setattr(vq, "_SLICED_ONE_LAUNCH", "0")
    for T in (2, 3, 4):
        sl.__dict__.pop(("_one_launch", T), None)
    y2g = m(xt[:, :2].contiguous())                # the gather kernel (this layer is too small for a launch per token)
    assert rel_err(tensor_to_bits(y2g), vo.forward(L, xs[:, :2]), "f16") <= 1e-3
    monkeypatch.setattr(vq, "_SLICED_ONE_LAUNCH", "auto")
    for T in (2, 3, 4):
        sl.__dict__.pop(("_one_launch", T), None)
    # v = 16 with a 1024-entry residual table: the gather kernel holds that table in LDS and is the faster one
    L16 = vo.make_layer(4096, 2048, dist="llm", seed=62, vector_len=16, num_centroids=65536, num_res_centroids=1024)
    m16 = spec_to_module(L16, dev)
    m16.enable_sliced_layout()
    m16(xt[:, :1].contiguous())
    sl16 = m16.__dict__["_sliced"][1]
    assert sl16 is not None and sl16.tokens_supported(2) and not m16._sliced_one_launch(sl16, 2)
    # inside a stream capture: the workspace exists (the calls above ran on this stream), the launch is captured
    g = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        m(xt[:, :2].contiguous())
        torch.cuda.synchronize()
        x2 = xt[:, :2].contiguous()
        with torch.cuda.graph(g, stream=st):
            yg = m(x2)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(yg.view(torch.int16), y2.view(torch.int16))


def test_module_default_route_for_two_to_four_tokens_of_wide_layers(dev, monkeypatch):
    """the product default (reference roundings), 2 - 4 tokens of v = 8 one-table layers whose exact layout has 16 slices: 2 / 3 tokens
    in ONE PASS of the one-token kernel where the operands fit its LDS (8192 columns), else - large layers - two launches (2 tokens)
    or the column-phase kernel (3 - 4 tokens) (VQuantLinear._sliced_one_launch / _sliced_token_limit: where each was measured
    faster than the gather kernel); everything else keeps the gather kernel; siblings share the launch"""
    import vptq_amd
    import vptq_amd.layers.vqlinear as vq
    from vptq_amd.layers.vqlinear import SiblingGroup
    assert vptq_amd.arithmetic() == "reference"
    L = vo.make_layer(8192, 6144, dist="llm", seed=71, num_centroids=65536, num_res_centroids=256, bias=True)
    m = spec_to_module(L, dev)
    xs = np.concatenate([_x(8192, "f16", "llm", 30 + i) for i in range(5)], axis=1)
    xt = bits_to_tensor(xs, "f16", dev).reshape(xs.shape)
    m(xt[:, :1].contiguous())
    sl = m.__dict__["_sliced"][1]
    assert sl is not None and sl.exact and sl.slices == 16 and all(m._sliced_one_launch(sl, t) for t in (2, 3, 4))
    assert sl.tokens_one_pass(2) and sl.tokens_one_pass(3) and not sl.tokens_one_pass(4) and not m._sliced_one_launch(sl, 5)
    want = vo.forward(L, xs)
    for T in (2, 3, 4):
        x = xt[:, :T].contiguous()
        y = m(x)
        assert torch.equal(y.view(torch.int16), sl.forward_tokens(x).view(torch.int16))
        yb = tensor_to_bits(y)
        assert rel_err(yb, want[:, :T], "f16") <= 1e-3
        assert float((yb.reshape(-1) == np.asarray(want[:, :T]).reshape(-1)).mean()) >= 0.95
    y5 = m(xt)                                     # 5 tokens: the gather kernel, the same arithmetic
    assert torch.equal(y5.view(torch.int16), gemv_abi(m, xt, EXACT).view(torch.int16))
    # a small 8192-column layer takes the one pass as well; a narrow layer (8 slices of 128 KiB) and a small 14336-column one (no room
    # for two tokens beside the slice, too small for the column-phase kernel to pay) stay on the gather kernel
    for (I, O, one) in ((8192, 512, True), (4096, 2048, False), (14336, 512, False)):
        Ls = vo.make_layer(I, O, dist="llm", seed=72, num_centroids=65536, num_res_centroids=0)
        ms = spec_to_module(Ls, dev)
        ms.enable_sliced_layout()
        x2 = bits_to_tensor(np.concatenate([_x(I, "f16", "llm", 40 + i) for i in range(2)], axis=1), "f16", dev).reshape(1, 2, I)
        ms(x2[:, :1].contiguous())
        s2 = ms.__dict__["_sliced"][1]
        assert s2 is not None and s2.exact and s2.tokens_supported(3) and ms._sliced_one_launch(s2, 2) == one and ms._sliced_token_limit(s2) == 1
        assert s2.tokens_window_parts(2) == (1 if I in (8192, 2048) else 2)
        assert torch.equal(ms(x2).view(torch.int16), (s2.forward_tokens(x2) if one else gemv_abi(ms, x2, EXACT)).view(torch.int16))
        assert rel_err(tensor_to_bits(ms(x2)), vo.forward(Ls, tensor_to_bits(x2)), "f16") <= 1e-3
    # a LARGE 14336-column layer: 2 / 3 tokens in one pass in two WINDOW PARTS (half of the columns' operands fit beside the slice);
    # 4 tokens: the column-phase kernel.  With the window parts switched off in the rule's eyes: two launches for 2 tokens
    Lw = vo.make_layer(14336, 3584, dist="llm", seed=73, num_centroids=65536, num_res_centroids=256)
    mw = spec_to_module(Lw, dev)
    xw = bits_to_tensor(np.concatenate([_x(14336, "f16", "llm", 50 + i) for i in range(4)], axis=1), "f16", dev).reshape(1, 4, 14336)
    mw(xw[:, :1].contiguous())
    sw = mw.__dict__["_sliced"][1]
    assert sw is not None and sw.exact and sw.tokens_window_parts(2) == 2 and sw.tokens_window_parts(3) == 2 and mw._sliced_token_limit(sw) == 2
    assert mw._sliced_one_launch(sw, 2) and mw._sliced_one_launch(sw, 3) and mw._sliced_one_launch(sw, 4)
    ww = vo.forward(Lw, tensor_to_bits(xw))
    for T in (2, 3, 4):
        x = xw[:, :T].contiguous()
        y = mw(x)
        assert torch.equal(y.view(torch.int16), sw.forward_tokens(x).view(torch.int16))
        yb = tensor_to_bits(y)
        assert rel_err(yb, ww[:, :T], "f16") <= 1e-3 and float((yb.reshape(-1) == np.asarray(ww[:, :T]).reshape(-1)).mean()) >= 0.95
    for T in (2, 3):   # (the one-token kernel over the same layout: the same weights; a row's sum is formed in two parts here)
        y = mw(xw[:, :T].contiguous())
        assert all(float((y[:, t].reshape(-1).view(torch.int16) == sw(xw[:, t:t + 1].contiguous()).reshape(-1).view(torch.int16)).float().mean()) >= 0.95
                   for t in range(T))
    # a 4096-column layer of that size (8 slices of 128 KiB): window parts too; a small one: the gather kernel
    Lg = vo.make_layer(4096, 12288, dist="llm", seed=75, num_centroids=65536, num_res_centroids=0, enable_perm=True)
    mg = spec_to_module(Lg, dev)
    xg = bits_to_tensor(np.concatenate([_x(4096, "f16", "llm", 60 + i) for i in range(3)], axis=1), "f16", dev).reshape(1, 3, 4096)
    mg(xg[:, :1].contiguous())
    sg = mg.__dict__["_sliced"][1]
    assert sg is not None and sg.exact and sg.slices == 8 and sg.tokens_window_parts(3) == 2 and mg._sliced_one_launch(sg, 3)
    yg = mg(xg)
    assert torch.equal(yg.view(torch.int16), sg.forward_tokens(xg).view(torch.int16))
    wg = vo.forward(Lg, tensor_to_bits(xg))
    assert rel_err(tensor_to_bits(yg), wg, "f16") <= 1e-3 and float((tensor_to_bits(yg).reshape(-1) == np.asarray(wg).reshape(-1)).mean()) >= 0.95
    # a two-table format of v = 8 ("4 bit"): 2 / 3 tokens in one pass as well - the residual entries are gathered from L2 once
    L4 = vo.make_layer(8192, 2048, dist="llm", seed=74, num_centroids=65536, num_res_centroids=65536)
    m4 = spec_to_module(L4, dev)
    m4(xt[:, :1].contiguous())
    s4 = m4.__dict__["_sliced"][1]
    assert s4 is not None and s4.exact and s4.res.dtype == torch.int16 and s4.tokens_one_pass(3) and not s4.tokens_supported(4)
    assert m4._sliced_one_launch(s4, 2) and m4._sliced_one_launch(s4, 3) and not m4._sliced_one_launch(s4, 4)
    for T in (2, 3):
        x = xt[:, :T].contiguous()
        y = m4(x)
        assert torch.equal(y.view(torch.int16), s4.forward_tokens(x).view(torch.int16))
        yb = tensor_to_bits(y)
        w4 = vo.forward(L4, xs[:, :T])
        assert rel_err(yb, w4, "f16") <= 1e-3 and float((yb.reshape(-1) == np.asarray(w4).reshape(-1)).mean()) >= 0.95
    assert torch.equal(m4(xt[:, :4].contiguous()).view(torch.int16), gemv_abi(m4, xt[:, :4].contiguous(), EXACT).view(torch.int16))
    # switched on for every layer the library takes ("1"): siblings of one format share one launch, each member's own bits
    monkeypatch.setattr(vq, "_SLICED_ONE_LAUNCH", "1")
    Lq = [vo.make_layer(8192, O, seed=80 + i, dist="llm", num_centroids=65536, num_res_centroids=256, bias=(i == 1)) for i, O in enumerate((1024, 264, 512))]
    mq = [spec_to_module(Li, dev) for Li in Lq]
    for mm in mq:
        mm.enable_sliced_layout()
    x3 = xt[:, :3].contiguous()
    alone = [mm(x3) for mm in mq]
    for mm in mq:
        assert mm.__dict__["_sliced"][1].exact and mm._sliced_one_launch(mm.__dict__["_sliced"][1], 3)
    group = SiblingGroup(mq)
    for mm in mq:
        object.__setattr__(mm, "_siblings", group)
    x3b = x3.clone()
    ys = [mm(x3b) for mm in mq]
    assert isinstance(group.__dict__.get("_sgroup"), tuple) and group._sgroup[1].exact
    for a, b, Li in zip(alone, ys, Lq):
        assert rel_err(tensor_to_bits(b), vo.forward(Li, xs[:, :3]), "f16") <= 1e-3
        assert float((a.view(torch.int16) == b.view(torch.int16)).float().mean()) >= 0.95   # (the group's rows per wave: another order of sums)


def test_sibling_layers_share_one_pass_in_window_parts(dev, monkeypatch):
    """q / k / v of a 4096-wide model, 3 tokens, reference roundings: ONE grouped launch of the one-pass kernel in two window parts
    (every member: half of the columns staged per workgroup), each member's result as from its own launch"""
    import vptq_amd
    import vptq_amd.layers.vqlinear as vq
    from vptq_amd.layers.vqlinear import SiblingGroup
    assert vptq_amd.arithmetic() == "reference"
    monkeypatch.setattr(vq, "_SLICED_ONE_LAUNCH", "1")       # (small test layers: below the auto rule's size)
    xs = np.concatenate([_x(4096, "f16", "llm", 90 + i) for i in range(3)], axis=1)
    x3 = bits_to_tensor(xs, "f16", dev).reshape(xs.shape)
    Lq = [vo.make_layer(4096, O, seed=95 + i, dist="llm", num_centroids=65536, num_res_centroids=256, bias=(i == 1)) for i, O in enumerate((1024, 264, 512))]
    mq = [spec_to_module(Li, dev) for Li in Lq]
    for mm in mq:
        mm.enable_sliced_layout()
    alone = [mm(x3) for mm in mq]
    for mm in mq:
        sq = mm.__dict__["_sliced"][1]
        assert sq.exact and sq.slices == 8 and sq.tokens_window_parts(3) == 2 and mm._sliced_one_launch(sq, 3)
    group = SiblingGroup(mq)
    for mm in mq:
        object.__setattr__(mm, "_siblings", group)
    x3b = x3.clone()
    ys = [mm(x3b) for mm in mq]
    assert isinstance(group.__dict__.get("_sgroup"), tuple) and group._sgroup[1].exact
    for a, b, Li in zip(alone, ys, Lq):
        assert rel_err(tensor_to_bits(a), vo.forward(Li, xs), "f16") <= 1e-3
        assert rel_err(tensor_to_bits(b), vo.forward(Li, xs), "f16") <= 1e-3
        assert float((a.view(torch.int16) == b.view(torch.int16)).float().mean()) >= 0.95   # (the group's rows per wave: another order of sums)


def test_module_route_for_two_and_three_tokens_over_column_parts(dev):
    """the product default: a 28672-column layer (two column parts of 14336) takes 2 / 3 tokens in one pass as well - every part in two
    window parts, 2 x 2 x 16 workgroups per row block meeting in the accumulator words; 4 tokens: the gather kernel"""
    L = vo.make_layer(28672, 2048, dist="llm", seed=76, num_centroids=65536, num_res_centroids=256, bias=True)
    m = spec_to_module(L, dev)
    xs = np.concatenate([_x(28672, "f16", "llm", 70 + i) for i in range(4)], axis=1)
    xt = bits_to_tensor(xs, "f16", dev).reshape(xs.shape)
    m(xt[:, :1].contiguous())
    sl = m.__dict__["_sliced"][1]
    assert sl is not None and sl.exact and sl.parts == 2 and sl.slices == 16 and sl.tokens_window_parts(2) == 2 and sl.tokens_window_parts(3) == 2
    assert m._sliced_one_launch(sl, 2) and m._sliced_one_launch(sl, 3) and not m._sliced_one_launch(sl, 4)
    want = vo.forward(L, xs)
    for T in (1, 2, 3):
        x = xt[:, :T].contiguous()
        y = m(x)
        if T > 1:
            assert torch.equal(y.view(torch.int16), sl.forward_tokens(x).view(torch.int16))
        yb = tensor_to_bits(y)
        assert rel_err(yb, want[:, :T], "f16") <= 1e-3, (T, rel_err(yb, want[:, :T], "f16"))
        assert float((yb.reshape(-1) == np.asarray(want[:, :T]).reshape(-1)).mean()) >= 0.95
    assert torch.equal(m(xt).view(torch.int16), gemv_abi(m, xt, EXACT).view(torch.int16))


@pytest.mark.parametrize("tokens", [2, 3, 4])
@pytest.mark.parametrize("v,kr,dt", [(8, 256, "f16"), (8, 0, "bf16"), (8, 65536, "f16"), (16, 65536, "f16"), (16, 0, "bf16")])
def test_sibling_layers_share_one_launch_for_two_to_four_tokens(v, kr, dt, tokens, dev, monkeypatch, folded_arithmetic):
    """q / k / v of a large-codebook model, 2 - 4 tokens: ONE launch of the token kernel for the group
    (`vptq_quant_gemv_sliced_tokens_grouped` via `SiblingGroup.forward_sliced`): bit-identical to the layers' own launches where the
    rows per wave do not change the order of a row's sums (they do not: a row is summed by one wave, phase by phase), one
    member with a permutation (its own pre-pass), one with an output bias"""
    import vptq_amd.layers.vqlinear as vq
    from vptq_amd.layers.vqlinear import SiblingGroup
    from vptq_amd.utils.sliced import SlicedGemv
    monkeypatch.setattr(vq, "_SLICED_ONE_LAUNCH", "1")       # (small test layers: below the auto rule's size)
    I = 2048
    outs = (1024, 33 * v, 512)
    Ls = [vo.make_layer(I, O, dist="llm", seed=170 + i + kr % 7 + tokens, dtype=dt, vector_len=v, num_centroids=65536, num_res_centroids=kr,
                        bias=(i == 1), enable_perm=(i == 2)) for i, O in enumerate(outs)]
    ms = [spec_to_module(L, dev) for L in Ls]
    x = _xt(I, tokens, dt, "llm", 31)
    xt = bits_to_tensor(x, dt, dev).reshape(x.shape)
    for m in ms:
        m.enable_sliced_layout()
    alone = [SlicedGemv(m).forward_tokens(xt) for m in ms]
    assert all(a is not None for a in alone)
    group = SiblingGroup(ms)
    for m in ms:
        object.__setattr__(m, "_siblings", group)
    [m(xt[:, :1].contiguous()) for m in ms]        # (one token first: builds the layouts and the group)
    ys = [m(xt) for m in ms]                       # the first call launches all three
    torch.cuda.synchronize()
    assert len(group._sout) == 0                   # every member picked its output up
    for L, y, a in zip(Ls, ys, alone):
        assert torch.equal(y.view(torch.int16), a.view(torch.int16))
        assert rel_err(tensor_to_bits(y), vo.forward(L, x), dt) <= TOL[dt]
    x2 = bits_to_tensor(_xt(I, tokens, dt, "llm", 32), dt, dev).reshape(x.shape)
    y2 = [m(x2) for m in ms]
    assert rel_err(tensor_to_bits(y2[2]), vo.forward(Ls[2], tensor_to_bits(x2)), dt) <= TOL[dt]


@pytest.mark.parametrize("tokens,dt", [(5, "f16"), (8, "f16"), (6, "bf16"), (8, "bf16")])
@pytest.mark.parametrize("v,k,kr", [(8, 65536, 0), (8, 65536, 256), (8, 65536, 65536), (16, 65536, 65536), (16, 65536, 0), (8, 65536, 1024)])
@pytest.mark.parametrize("I,O,kw", [(2048, 528, dict(dist="llm", enable_perm=True, bias=True)), (4096, 272, dict(dist="llm", bias=True)),
                                    (4104, 136, dict(dist="llm")), (72, 1040, dict())])
def test_sliced_tokens_five_to_eight(I, O, kw, v, k, kr, tokens, dt, dev):
    """5 - 8 tokens in one launch: 8 token slots on the matrix pipe (16 bytes of activations per column: layers of up to ~4600
    columns), against the oracle; float32 outputs; repeatable"""
    from vptq_amd.utils.sliced import SlicedGemv
    from vptq_amd import _backend as B
    kw = dict(kw)
    dist = kw.pop("dist", "ref-test")
    L = vo.make_layer(I, O, dist=dist, seed=I + O + v + kr + tokens, dtype=dt, vector_len=v, num_centroids=k, num_res_centroids=kr, **kw)
    x = _xt(I, tokens, dt, dist, I + 5)
    m = spec_to_module(L, dev)
    sl = SlicedGemv(m)
    xt = bits_to_tensor(x, dt, dev).reshape(x.shape)
    assert sl.tokens_supported(tokens)
    got = sl.forward_tokens(xt)
    torch.cuda.synchronize()
    assert got.shape == (1, tokens, O)
    err = rel_err(tensor_to_bits(got), vo.forward(L, x), dt)
    assert err <= TOL[dt], f"v{v}-k{k}-{kr} {I}x{O} {tokens} tokens {dt}: {err:.3e}"
    y32 = sl.forward_tokens(xt, flags=B.GEMV_OUT_F32)
    assert torch.equal(y32.to(got.dtype).view(torch.int16), got.view(torch.int16))
    assert torch.equal(sl.forward_tokens(xt).view(torch.int16), got.view(torch.int16))
    # the 4-token workspace was replaced by the larger one: fewer tokens still run in it
    assert rel_err(tensor_to_bits(sl.forward_tokens(xt[:, :3].contiguous())), vo.forward(L, x[:, :3]), dt) <= TOL[dt]


def test_sliced_tokens_five_to_eight_need_narrow_layers(dev):
    from vptq_amd.utils.sliced import SlicedGemv
    L = vo.make_layer(8192, 64, dist="llm", seed=91, num_centroids=65536, num_res_centroids=0)
    sl = SlicedGemv(spec_to_module(L, dev))
    assert sl.tokens_supported(4) and not sl.tokens_supported(5) and not sl.tokens_supported(9)
    assert sl.forward_tokens(torch.randn(1, 6, 8192, device=dev, dtype=torch.float16)) is None
