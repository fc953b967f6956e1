"""GPU parity: the HIP path (through the C ABI) against the oracle and the golden
vectors produced by the real reference.  Run with `pytest -m gpu` on an MI355X.

Tolerances: outputs within 1e-3 of the reference CPU path,
measured as max|d| / max|ref| for fp16 (1 fp16 ulp ~ 4.9e-4); bf16 has 8 bits of
mantissa (ulp 3.9e-3) so its bar is 8e-3.  Dequantised weights are bit-exact.
"""
import hashlib
import os

import numpy as np
import pytest
import torch

from oracle import vptq_oracle as vo
from _cases import golden_names, load_golden, rel_err, bit_identical_frac
from _gpu_util import (spec_to_module, bits_to_tensor, tensor_to_bits, gemv_abi, kernel_name,
                       module_desc, module_to_spec, TORCH_DT)

pytestmark = pytest.mark.gpu
TOL = {"f16": 1e-3, "bf16": 8e-3}
FAST, GENERIC, EXACT, MFMA, VALU = 1, 2, 4, 8, 16


def module_flags():
    """the flags the module forward passes to the library for a layer with norm tensors: the reference's roundings
    (VPTQ_GEMV_EXACT) unless the folded arithmetic is opted in (vptq_amd.set_arithmetic("folded"); the `folded_arithmetic`
    fixture) - and then still for the layers its measured gate turns down; VPTQ_EXACT=1 in the environment forces them"""
    import vptq_amd
    from vptq_amd import ops
    return ops.quant_gemm_flags() | (EXACT if vptq_amd.arithmetic() == "reference" else 0)


def expect_kernel(m, tokens, flags, name):
    """the library's kernel choice - unless VPTQ_K256_KERNEL pins it for an A/B run"""
    if os.environ.get("VPTQ_K256_KERNEL"):
        return
    assert kernel_name(m, tokens, flags) == name, (kernel_name(m, tokens, flags), name)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from vptq_amd import _backend as B
    B.lib()  # fail loudly if libvptq_hip.so is missing
    return torch.device("cuda", 0)


# ---------------------------------------------------------------- golden vectors
@pytest.mark.parametrize("name", golden_names())
def test_dequant_bit_exact_vs_reference_golden(name, dev):
    L, x, y, cfg, W_head = load_golden(name)
    m = spec_to_module(L, dev)
    W = tensor_to_bits(m.dequant())
    assert W.shape == (cfg["out_features"], cfg["in_features"])
    assert (W[:16] == W_head).all()
    assert hashlib.sha256(W.tobytes()).hexdigest() == cfg["W_sha256"]


@pytest.mark.parametrize("name", golden_names())
def test_forward_vs_reference_golden(name, dev):
    L, x, y, cfg, _ = load_golden(name)
    dt = cfg["dtype"]
    m = spec_to_module(L, dev)
    xt = bits_to_tensor(x, dt, dev).reshape(x.shape)
    out = tensor_to_bits(m(xt))
    assert out.shape == y.shape
    err = rel_err(out, y, dt)
    assert err <= TOL[dt], f"{name}: {err:.3e}"
    # against the oracle too (same numbers, different summation order)
    assert rel_err(out, vo.forward(L, x), dt) <= TOL[dt]


@pytest.mark.parametrize("name", [n for n in golden_names() if n.startswith("canon")])
def test_canonical_layers_take_the_specialised_kernel(name, dev):
    L, x, y, cfg, _ = load_golden(name)
    m = spec_to_module(L, dev)
    assert kernel_name(m, cfg["tokens"]).startswith("gemv_k256" if cfg["tokens"] <= 4 else "gemm_k256")
    assert kernel_name(m, cfg["tokens"], GENERIC) == "gemv_generic_kernel"
    xt = bits_to_tensor(x, cfg["dtype"], dev).reshape(x.shape)
    a = tensor_to_bits(gemv_abi(m, xt, EXACT))
    b = tensor_to_bits(gemv_abi(m, xt, GENERIC))
    # both rebuild identical weights; only the fp32 summation order differs
    assert rel_err(a, b, cfg["dtype"]) <= TOL[cfg["dtype"]] / 2
    assert bit_identical_frac(a, b) > 0.75, bit_identical_frac(a, b)
    for out in (a, b):
        assert rel_err(out, y, cfg["dtype"]) <= TOL[cfg["dtype"]]


# ---------------------------------------------------------------- seeded layers vs oracle
CASES = [
    # I, O, kwargs, tokens
    (1024, 512, dict(), 1),
    (1024, 512, dict(enable_perm=True, bias=True), 2),
    (4096, 264, dict(), 1),                       # one full sweep, rows not a multiple of 4
    (4104, 64, dict(), 3),                        # G not a multiple of the sweep (tail lanes)
    (8192 + 512, 40, dict(dist="llm"), 1),        # > one iteration of two sweeps
    (512, 1000, dict(dtype="bf16", dist="llm"), 1),
    (768, 136, dict(num_centroids=4096, num_res_centroids=0, vector_len=6), 4),
    (640, 250, dict(num_centroids=65536, num_res_centroids=256, dist="llm"), 1),   # T=24, padding
    (512, 96, dict(num_centroids=8192, num_res_centroids=4096, vector_len=12, dist="llm"), 8),
    (16 + 2 * 256, 100, dict(num_codebooks=2, outlier_size=16, outlier_vector_len=4,
                             num_outlier_centroids=64, enable_perm=True, bias=True), 1),
    (256, 64, dict(enable_norm=False, num_res_centroids=0, num_centroids=16, vector_len=2), 1),
    (1024, 256, dict(vector_len=16, num_centroids=65536, num_res_centroids=65536, dist="llm"), 2),
    # large-codebook v=8 formats -> gemv_gather_kernel (T = 16 / 24 / 32)
    (2048, 512, dict(num_centroids=65536, num_res_centroids=0, dist="llm"), 1),
    (2056, 136, dict(num_centroids=65536, num_res_centroids=0, dist="llm", enable_perm=True, bias=True), 3),
    (1024, 2048 * 8, dict(num_centroids=65536, num_res_centroids=256, dist="llm"), 1),      # ROWS = 2
    (1032, 264, dict(num_centroids=65536, num_res_centroids=256, enable_perm=True), 2),
    (1024, 256, dict(num_centroids=65536, num_res_centroids=65536, dist="llm"), 4),
    (512, 96, dict(num_centroids=65536, num_res_centroids=65536, dtype="bf16", dist="llm"), 7),
    # 5-8 tokens: one pass over the indices with 8 token slots; 9-16: two
    (2048, 264, dict(num_centroids=65536, num_res_centroids=256, dist="llm", bias=True), 5),
    (1032, 512, dict(num_centroids=65536, num_res_centroids=0, enable_perm=True), 8),
    (1024, 128, dict(num_centroids=65536, num_res_centroids=65536, dist="llm"), 6),
    (776, 200, dict(num_centroids=65536, num_res_centroids=256, dtype="bf16", dist="llm", enable_perm=True), 8),
    (1024, 256, dict(num_centroids=65536, num_res_centroids=256, dist="llm"), 13),
]


@pytest.mark.parametrize("I,O,kw,tokens", CASES)
def test_gemv_and_dequant_vs_oracle(I, O, kw, tokens, dev):
    kw = dict(kw)
    dist = kw.pop("dist", "ref-test")
    L = vo.make_layer(I, O, dist=dist, seed=I + O, **kw)
    dt = L.dtype
    rng = np.random.default_rng(5)
    xs = (0.02 + 0.5 * rng.standard_normal((1, tokens, I))) if dist == "ref-test" \
        else rng.standard_normal((1, tokens, I))
    x = vo.from_f32(xs.astype(np.float32), dt)
    m = spec_to_module(L, dev)
    if L.num_centroids == 65536 and L.vector_len == 8 and L.num_res_centroids in (0, 256, 65536):
        assert kernel_name(m, tokens) == "gemv_gather_kernel"
    W_ref = vo.dequant(L, ref_residual_mask_quirk=False)
    assert (tensor_to_bits(m.dequant()) == W_ref).all(), "dequant must be bit-exact"
    want = vo.gemv(W_ref, x, dt, L.bias)
    xt = bits_to_tensor(x, dt, dev).reshape(x.shape)
    got = tensor_to_bits(m(xt))
    assert got.shape == want.shape
    err = rel_err(got, want, dt)
    assert err <= TOL[dt], f"{err:.3e}"
    if tokens > 4:   # the fused route through the C ABI whatever route the module takes at this count
        assert rel_err(tensor_to_bits(gemv_abi(m, xt, module_flags())), want, dt) <= TOL[dt]


@pytest.mark.parametrize("k", [256, 65536])
@pytest.mark.parametrize("tokens", [1, 2, 3, 4, 5, 8, 9, 13, 16, 17, 33, 49])
def test_token_counts_gemv_and_gemm_paths(tokens, k, dev):
    """1..8 tokens (canonical 256 + 256 format: 1..48, from 5 on the batched-decode kernel in launches
    of 16): fused; more: dequant + F.linear (the reference switches at 3).  Module forward and the
    functional op agree."""
    from vptq_amd import _backend as B
    kw = dict(num_res_centroids=256) if k == 256 else dict(num_centroids=65536, num_res_centroids=-1)
    L = vo.make_layer(1024, 256, dist="llm", seed=11, bias=True, **kw)
    x = vo.from_f32(np.random.default_rng(tokens).standard_normal((2, tokens, 1024))
                    .astype(np.float32), "f16")[:1]
    m = spec_to_module(L, dev)
    desc, keep = module_desc(m)
    assert B.lib().vptq_quant_gemv_max_tokens(desc) == (48 if k == 256 else 8)
    xt = bits_to_tensor(x, "f16", dev).reshape(x.shape)
    got = tensor_to_bits(m(xt))
    want = vo.forward(L, x)
    assert rel_err(got, want, "f16") <= 1e-3
    if tokens <= 16 or k == 256:   # the C ABI accepts up to 16 tokens for every format (canonical fp16: 64)
        flags = module_flags()   # what the module forward passes (VPTQ_EXACT=1: exact)
        assert rel_err(tensor_to_bits(gemv_abi(m, xt, flags)), want, "f16") <= 1e-3
        if tokens <= (48 if k == 256 else 8):
            if k == 65536 and tokens == 1:
                # (round 3: one token of a k = 65536 layer goes over the load-time derived sliced layout by default -
                # another kernel, same results within the bar; the gather kernel's bits with it switched off)
                m.enable_sliced_layout(False)
                got = tensor_to_bits(m(xt))
            assert (tensor_to_bits(gemv_abi(m, xt, flags)) == got).all()   # forward took the fused path


def test_default_and_exact_arithmetic(dev):
    """flags = 0 is the folded fp32 form (inside the 1e-3 bar); VPTQ_GEMV_EXACT rebuilds the
    reference's rounded weights (almost every output bit-identical to the oracle)."""
    for dist in ("ref-test", "llm"):
        L = vo.make_layer(4096, 512, dist=dist, seed=3)
        x = vo.from_f32((np.random.default_rng(1).standard_normal((1, 1, 4096)) *
                         (0.5 if dist == "ref-test" else 1.0)).astype(np.float32), "f16")
        m = spec_to_module(L, dev)
        xt = bits_to_tensor(x, "f16", dev).reshape(x.shape)
        want = vo.forward(L, x)
        expect_kernel(m, 1, 0, "gemv_k256_kernel<fast>")
        expect_kernel(m, 1, EXACT, "gemv_k256_kernel")
        exact = tensor_to_bits(gemv_abi(m, xt, EXACT))
        dflt = tensor_to_bits(gemv_abi(m, xt, 0))
        assert rel_err(exact, want, "f16") <= 2.5e-4       # rounding-exact weights
        assert rel_err(dflt, want, "f16") <= 1e-3          # folded fp32 form
        assert bit_identical_frac(exact, want) >= 0.95
        # FAST_MATH (ABI 2's opt-in) = the default now; the module forward uses the default
        assert (tensor_to_bits(gemv_abi(m, xt, FAST)) == dflt).all()
        # (the reference test's distribution is above the load-time gate's bias line: the module serves it exact)
        assert (tensor_to_bits(m(xt)) == (exact if (module_flags() | m._descriptor()[9]) & EXACT else dflt)).all()


@pytest.mark.parametrize("name", [n for n in golden_names() if n.startswith("canon")])
def test_canonical_goldens_through_both_kernels(name, dev):
    """The real reference's outputs (1, 2, 4 and 16 tokens, fp16 / bf16, perm, bias) against the
    persistent MFMA kernel and the VALU kernel, each in both arithmetic forms."""
    L, x, y, cfg, _ = load_golden(name)
    dt = cfg["dtype"]
    m = spec_to_module(L, dev)
    xt = bits_to_tensor(x, dt, dev).reshape(x.shape)
    for flags in (MFMA, VALU, MFMA | EXACT, VALU | EXACT):
        out = tensor_to_bits(gemv_abi(m, xt, flags))
        assert out.shape == y.shape
        err = rel_err(out, y, dt)
        assert err <= TOL[dt], f"{name} flags={flags}: {err:.3e}"
    if cfg["tokens"] <= 4:
        expect_kernel(m, cfg["tokens"], MFMA, "gemv_k256m_kernel<fast>")


# I, O, kwargs: the persistent MFMA kernel, forced onto small layers: every sweep count
# 1..7 (I / 2048), row counts that are no multiple of 4, more row groups than CUs, the input
# permutation, an output bias
MFMA_CASES = [
    (2048, 512, dict()),
    (1024, 40, dict(bias=True)),                      # 5 vector-rows: a partial row group
    (4096, 264, dict(dist="llm")),
    (4104, 64, dict()),                               # 3 sweeps, tail lanes past G
    (8192, 96, dict(dist="llm")),
    (8192 + 512, 40, dict(dist="llm")),               # 5 sweeps, two staging blocks
    (11008, 64, dict(dist="llm")),
    (14336, 72, dict(dist="llm", bias=True)),         # 7 sweeps: the widest layer staged in LDS
    (16384, 40, dict(dist="llm")),                    # wider: column blocks, x through the queue
    (28672, 72, dict(dist="llm", bias=True)),         # Llama-3-70B down_proj width
    (20480 + 8, 136, dict(dist="llm")),               # last column block almost empty
    (16384, 32, dict(enable_perm=True)),              # wide + permutation: the VALU kernel
    (1024, 8 * 4 * 300, dict(dist="llm")),            # 300 row groups on 256 CUs
    (2048, 520, dict(enable_perm=True, bias=True)),
    (8192 + 8, 136, dict(enable_perm=True, dist="llm")),
]


@pytest.mark.parametrize("I,O,kw", MFMA_CASES)
def test_mfma_kernel_vs_oracle(I, O, kw, dev):
    kw = dict(kw)
    dist = kw.pop("dist", "ref-test")
    L = vo.make_layer(I, O, dist=dist, seed=I + O, **kw)
    rng = np.random.default_rng(6)
    xs = (0.02 + 0.5 * rng.standard_normal((1, 1, I))) if dist == "ref-test" \
        else rng.standard_normal((1, 1, I))
    x = vo.from_f32(xs.astype(np.float32), "f16")
    m = spec_to_module(L, dev)
    xt = bits_to_tensor(x, "f16", dev).reshape(x.shape)
    want = vo.forward(L, x)
    wide_perm = I > 14336 and kw.get("enable_perm", False)
    expect_kernel(m, 1, MFMA, "gemv_k256_kernel<fast>" if wide_perm else "gemv_k256m_kernel<fast>")
    expect_kernel(m, 1, VALU, "gemv_k256_kernel<fast>")
    got = tensor_to_bits(gemv_abi(m, xt, MFMA))
    assert rel_err(got, want, "f16") <= 1e-3
    # same arithmetic as the VALU kernel up to summation order and one f16 rounding of s * x
    assert rel_err(got, tensor_to_bits(gemv_abi(m, xt, VALU)), "f16") <= 1e-3
    # exact form on the matrix pipe (every width staged in LDS; 6 and 7 sweeps with scale and bias staged
    # beside the activations - round 6): bit-identical weights, fp32 sums in another order
    expect_kernel(m, 1, MFMA | EXACT, "gemv_k256m_kernel" if I <= 14336 else "gemv_k256_kernel")
    ex = tensor_to_bits(gemv_abi(m, xt, MFMA | EXACT))
    assert rel_err(ex, want, "f16") <= 5e-4      # same weights; a flipped last bit at most
    assert bit_identical_frac(ex, want) >= 0.9


@pytest.mark.parametrize("I,O", [(8192, 8192), (4096, 14336), (2048, 9600), (28672, 2048),
                                 (1024, 49152),      # 6 row groups per workgroup: slot reuse
                                 (14336, 12288)])    # one partial-sum slot (K = 1): waves wait
def test_mfma_kernel_is_deterministic(I, O, dev):
    """The persistent kernel hands partial sums from wave to wave through LDS counters and streams
    its index words through a register queue: 25 launches over the same inputs (1 to 4 row
    groups per workgroup) must give the same bits every time, in both arithmetic forms."""
    L = vo.make_layer(I, O, dist="llm", seed=I ^ O)
    m = spec_to_module(L, dev)
    x = bits_to_tensor(vo.from_f32(np.random.default_rng(9).standard_normal((1, 1, I))
                                   .astype(np.float32), "f16"), "f16", dev).reshape(1, 1, I)
    for flags in (MFMA, MFMA | EXACT):
        first = gemv_abi(m, x, flags)
        for _ in range(24):
            assert torch.equal(gemv_abi(m, x, flags), first)
        ref = gemv_abi(m, x, (flags & EXACT) | VALU)
        assert rel_err(tensor_to_bits(first), tensor_to_bits(ref), "f16") <= 1e-3


@pytest.mark.parametrize("I,O,kw", [
    (2048, 512, dict()),
    (4096, 1024, dict(dist="llm", bias=True)),        # 32 row groups: bf16 already takes it
    (8192 + 512, 264, dict(dist="llm")),
    (4104, 72, dict(enable_perm=True)),
    (14336, 256, dict(dist="llm")),                   # 7 sweeps: scale and bias still fit beside the activations
    (2048, 4608, dict(dist="llm")),                   # 144 row groups
    (16384, 64, dict(dist="llm")),                    # wider than the LDS can stage
])
def test_mfma_kernel_bf16(I, O, kw, dev):
    """bf16 goes through the MFMA kernel's folded form (the VALU kernel only has the widened
    exact arithmetic for bf16); VPTQ_GEMV_EXACT and FORCE_VALU fall back to that."""
    kw = dict(kw)
    dist = kw.pop("dist", "ref-test")
    L = vo.make_layer(I, O, dist=dist, seed=I + O, dtype="bf16", **kw)
    rng = np.random.default_rng(8)
    xs = (0.02 + 0.5 * rng.standard_normal((1, 1, I))) if dist == "ref-test" \
        else rng.standard_normal((1, 1, I))
    x = vo.from_f32(xs.astype(np.float32), "bf16")
    m = spec_to_module(L, dev)
    xt = bits_to_tensor(x, "bf16", dev).reshape(x.shape)
    want = vo.forward(L, x)
    expect_kernel(m, 1, MFMA, "gemv_k256m_kernel<fast>")
    # the reference's roundings: on the matrix pipe (round 6) while scale and bias fit into LDS beside the activations
    exact_mfma = I <= 14336
    expect_kernel(m, 1, MFMA | EXACT, "gemv_k256m_kernel" if exact_mfma else "gemv_k256_kernel")
    expect_kernel(m, 2, MFMA | EXACT, "gemv_k256m_kernel" if exact_tokens_on_matrix_pipe(I, 2, "bf16") else "gemv_k256_kernel")
    expect_kernel(m, 2, EXACT, "gemv_k256m_kernel" if exact_tokens_on_matrix_pipe(I, 2, "bf16") and (O + 31) // 32 >= 144 else "gemv_k256_kernel")
    expect_kernel(m, 1, EXACT, "gemv_k256m_kernel" if exact_mfma and (O + 31) // 32 >= 32 else "gemv_k256_kernel")
    expect_kernel(m, 1, VALU, "gemv_k256_kernel")
    expect_kernel(m, 1, 0, "gemv_k256m_kernel<fast>" if (O + 31) // 32 >= 32 else "gemv_k256_kernel")
    expect_kernel(m, 2, 0, "gemv_k256m_kernel<fast>" if (O + 31) // 32 >= 32 else "gemv_k256_kernel")
    got = tensor_to_bits(gemv_abi(m, xt, MFMA))
    assert rel_err(got, want, "bf16") <= TOL["bf16"]
    assert rel_err(tensor_to_bits(gemv_abi(m, xt, VALU)), want, "bf16") <= TOL["bf16"]
    assert rel_err(tensor_to_bits(gemv_abi(m, xt, MFMA | EXACT)), want, "bf16") <= TOL["bf16"]
    # both kernels round every weight like the reference: their fp32 outputs differ by the order of the fp32 sums only
    ye, yv = gemv_abi(m, xt, MFMA | EXACT, out_f32=True).double(), gemv_abi(m, xt, VALU | EXACT, out_f32=True).double()
    assert float((ye - yv).abs().max()) <= 2e-6 * float(yv.abs().max())
    assert (tensor_to_bits(m(xt)) == tensor_to_bits(gemv_abi(m, xt, module_flags()))).all()
    first = gemv_abi(m, xt, MFMA)
    for _ in range(5):
        assert torch.equal(gemv_abi(m, xt, MFMA), first)


MFMA_TOKEN_CASES = [
    # I, O, kwargs, tokens
    (2048, 512, dict(), 2),
    (1024, 40, dict(bias=True), 3),                   # 5 vector-rows: a partial row group
    (4104, 64, dict(), 4),                            # tail lanes past G
    (8192, 96, dict(dist="llm"), 2),
    (8192, 264, dict(dist="llm", bias=True), 4),
    (8192 + 512, 40, dict(dist="llm"), 3),            # two staging blocks
    (11008, 64, dict(dist="llm"), 4),                 # 4 token slots: the widest that fit (1 slot)
    (14336, 72, dict(dist="llm"), 2),                 # 2 token slots, 7 sweeps
    (1024, 8 * 4 * 300, dict(dist="llm"), 4),         # 300 row groups on 256 CUs
    (2048, 520, dict(enable_perm=True, bias=True), 2),
    (8192 + 8, 136, dict(enable_perm=True, dist="llm"), 4),
]


def exact_tokens_on_matrix_pipe(I, tokens, dt):
    """where the persistent kernel serves 2 - 4 tokens in the reference's roundings (round 6; gemv_k256m_supported): 2 token slots
    at every width whose operands fit into LDS (activations per slot + scale + bias planes), 4 slots up to 2 (fp16) / 1 (bf16)
    sweeps of 2048 columns (beyond: spills)"""
    slots = 2 if tokens == 2 else 4
    if slots == 4 and -(-I // 2048) > (2 if dt == "f16" else 1):
        return False
    lds = 65536 + (slots + 2) * (2 * I + 32) + slots * 64 + 64 + slots * 2048
    return lds <= 160 * 1024


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("I,O,kw,tokens", MFMA_TOKEN_CASES)
def test_mfma_kernel_several_tokens(I, O, kw, tokens, dt, dev):
    """2-4 activation rows per launch: the MFMA kernel's contraction form (token = row of the
    4x4x4 block), folded arithmetic, against the oracle and against the VALU kernel; every
    token row must equal the one-token launch of that row up to the summation order."""
    kw = dict(kw)
    dist = kw.pop("dist", "ref-test")
    L = vo.make_layer(I, O, dist=dist, seed=I + O + tokens, dtype=dt, **kw)
    rng = np.random.default_rng(tokens)
    xs = (0.02 + 0.5 * rng.standard_normal((1, tokens, I))) if dist == "ref-test" \
        else rng.standard_normal((1, tokens, I))
    x = vo.from_f32(xs.astype(np.float32), dt)
    m = spec_to_module(L, dev)
    xt = bits_to_tensor(x, dt, dev).reshape(x.shape)
    want = vo.forward(L, x)
    expect_kernel(m, tokens, MFMA, "gemv_k256m_kernel<fast>")
    # the reference's roundings: the one-token exact loop with a pair of MFMAs more per token, where it fits (round 6)
    on_pipe = exact_tokens_on_matrix_pipe(I, tokens, dt)
    expect_kernel(m, tokens, MFMA | EXACT, "gemv_k256m_kernel" if on_pipe else "gemv_k256_kernel")
    ex = gemv_abi(m, xt, MFMA | EXACT)
    assert rel_err(tensor_to_bits(ex), want, dt) <= (5e-4 if dt == "f16" else TOL[dt])
    assert bit_identical_frac(tensor_to_bits(ex), want) >= 0.9
    # same weights as the VALU kernel's exact form: fp32 outputs differ by the order of the fp32 sums only
    ye, yv = gemv_abi(m, xt, MFMA | EXACT, out_f32=True).double(), gemv_abi(m, xt, VALU | EXACT, out_f32=True).double()
    assert float((ye - yv).abs().max()) <= 2e-6 * float(yv.abs().max())
    for _ in range(3):
        assert torch.equal(gemv_abi(m, xt, MFMA | EXACT), ex)
    got_t = gemv_abi(m, xt, MFMA)
    got = tensor_to_bits(got_t)
    assert got.shape == want.shape
    assert rel_err(got, want, dt) <= TOL[dt]
    assert rel_err(got, tensor_to_bits(gemv_abi(m, xt, VALU)), dt) <= TOL[dt]
    for t in range(tokens):
        one = tensor_to_bits(gemv_abi(m, xt[:, t:t + 1].contiguous(), MFMA))
        assert rel_err(got[:, t:t + 1], one, dt) <= TOL[dt]   # summation order: an ulp at most
    for _ in range(5):
        assert torch.equal(gemv_abi(m, xt, MFMA), got_t)


def test_mfma_kernel_several_tokens_limits(dev):
    """Where the token slots do not fit beside the codebook image the VALU kernel serves."""
    Lw = vo.make_layer(14336, 64, dist="llm", seed=5)
    wide = spec_to_module(Lw, dev)
    expect_kernel(wide, 2, MFMA, "gemv_k256m_kernel<fast>")
    expect_kernel(wide, 4, MFMA, "gemv_k256_kernel")              # 4 x 28 KiB of activations
    wider = spec_to_module(vo.make_layer(16384, 64, dist="llm", seed=6), dev)
    expect_kernel(wider, 1, MFMA, "gemv_k256m_kernel<fast>")      # unstaged: one token only
    expect_kernel(wider, 2, MFMA, "gemv_k256_kernel<fast>")
    x = vo.from_f32(np.random.default_rng(4).standard_normal((1, 4, 14336)).astype(np.float32), "f16")
    got = tensor_to_bits(gemv_abi(wide, bits_to_tensor(x, "f16", dev).reshape(x.shape), MFMA))
    assert rel_err(got, vo.forward(Lw, x), "f16") <= 1e-3


def test_mfma_kernel_is_the_default_for_large_launches(dev):
    """From 144 row groups of 4 vector-rows on (where the VALU kernel needs a second round of
    workgroups), in both arithmetic forms, one token."""
    L = vo.make_layer(1024, 8192, dist="llm", seed=77)
    m = spec_to_module(L, dev)
    expect_kernel(m, 1, 0, "gemv_k256m_kernel<fast>")
    expect_kernel(m, 1, EXACT, "gemv_k256m_kernel")
    expect_kernel(m, 2, 0, "gemv_k256m_kernel<fast>")
    expect_kernel(m, 4, 0, "gemv_k256m_kernel<fast>")
    expect_kernel(m, 2, EXACT, "gemv_k256m_kernel")            # several tokens in the reference's roundings (round 6)
    expect_kernel(m, 4, EXACT, "gemv_k256m_kernel")            # (1 sweep: the 4-slot instantiation fits)
    small = spec_to_module(vo.make_layer(1024, 4096, dist="llm", seed=78), dev)
    expect_kernel(small, 1, 0, "gemv_k256_kernel<fast>")
    x = vo.from_f32(np.random.default_rng(3).standard_normal((1, 1, 1024)).astype(np.float32), "f16")
    got = tensor_to_bits(m(bits_to_tensor(x, "f16", dev).reshape(x.shape)))
    assert rel_err(got, vo.forward(L, x), "f16") <= 1e-3


def test_edge_index_patterns(dev):
    """all-zero / all-ones streams, cyclic arange (the reference test's pattern)."""
    I, O = 1024, 512
    for pattern in ("zeros", "ones", "cyclic"):
        L = vo.make_layer(I, O, dist="llm", seed=9)
        if pattern == "zeros":
            L.indices = np.zeros_like(L.indices)
        elif pattern == "ones":
            L.indices = np.full_like(L.indices, -1)
        else:
            idx = (np.arange(L.num_indices * I) % 256).reshape(1, L.num_indices, I)
            L.indices = vo.pack_indices(idx, 8, (idx * 7 + 3) % 256, 8)
        x = vo.from_f32(np.random.default_rng(2).standard_normal((1, 1, I)).astype(np.float32), "f16")
        m = spec_to_module(L, dev)
        assert (tensor_to_bits(m.dequant()) == vo.dequant(L)).all()
        got = tensor_to_bits(m(bits_to_tensor(x, "f16", dev).reshape(x.shape)))
        assert rel_err(got, vo.forward(L, x), "f16") <= 1e-3


def test_empty_and_extreme_inputs(dev):
    """zero tokens; the widest layer a uint16 permutation allows (I = 65536); the largest
    index values of a 65536-entry codebook."""
    L = vo.make_layer(1024, 256, dist="llm", seed=3, bias=True)
    m = spec_to_module(L, dev)
    y = m(torch.empty(2, 0, 1024, device=dev, dtype=torch.float16))
    assert tuple(y.shape) == (2, 0, 256)
    # I = 65536 with perm (max for uint16), 2 tokens, padded O
    L = vo.make_layer(65536, 60, dist="llm", seed=4, enable_perm=True)
    x = vo.from_f32(np.random.default_rng(4).standard_normal((1, 2, 65536)).astype(np.float32), "f16")
    m = spec_to_module(L, dev)
    got = tensor_to_bits(m(bits_to_tensor(x, "f16", dev).reshape(x.shape)))
    assert rel_err(got, vo.forward(L, x), "f16") <= 1e-3
    # k = 65536: every index = 65535 / residual = 255
    L = vo.make_layer(512, 128, dist="llm", seed=5, num_centroids=65536, num_res_centroids=256)
    L.indices = np.full_like(L.indices, -1)
    x = vo.from_f32(np.random.default_rng(5).standard_normal((1, 1, 512)).astype(np.float32), "f16")
    m = spec_to_module(L, dev)
    assert (tensor_to_bits(m.dequant()) == vo.dequant(L)).all()
    got = tensor_to_bits(m(bits_to_tensor(x, "f16", dev).reshape(x.shape)))
    assert rel_err(got, vo.forward(L, x), "f16") <= 1e-3


# ---------------------------------------------------------------- BASELINE sizes
@pytest.mark.parametrize("H", [4096, 8192])
def test_full_size_layers_properties(H, dev):
    """At BASELINE.json sizes the numpy oracle is slow, so: (a) oracle on a random
    subset of vector-rows, (b) HIP GEMV == HIP dequant followed by an fp32 matmul
    (two independent kernels), (c) specialised == generic kernel."""
    L = vo.make_layer(H, H, dist="llm", seed=H)
    rng = np.random.default_rng(7)
    x = vo.from_f32(rng.standard_normal((1, 1, H)).astype(np.float32), "f16")
    m = spec_to_module(L, dev)
    assert kernel_name(m, 1).startswith("gemv_k256")
    xt = bits_to_tensor(x, "f16", dev).reshape(x.shape)
    y = m(xt)
    got = tensor_to_bits(y)
    # (a) oracle on 48 random vector-rows
    rows = np.sort(rng.choice(L.num_indices, 48, replace=False))
    sub = vo.LayerSpec(H, 48 * 8, 8, 256, 256, 1, H, dtype="f16")
    sub.indices = L.indices[:, rows, :]
    sub.centroids, sub.res_centroids = L.centroids, L.res_centroids
    sub.weight_scale, sub.weight_bias = L.weight_scale, L.weight_bias
    want = vo.forward(sub, x).reshape(48, 8)
    mine = got.reshape(-1, 8)[rows]
    scale = np.abs(vo.to_f32(got, "f16")).max()
    d = np.abs(vo.to_f32(mine, "f16") - vo.to_f32(want, "f16")).max()
    assert d / scale <= 1e-3
    # (b) dequant + fp32 matmul on the GPU
    W = m.dequant()
    ref = (xt.float().reshape(1, H) @ W.float().t()).reshape(-1)
    assert ((y.float().reshape(-1) - ref).abs().max() / ref.abs().max()).item() <= 1e-3
    # (c) generic kernel
    g = gemv_abi(m, xt, GENERIC)
    assert ((y.float() - g.float()).abs().max() / ref.abs().max()).item() <= 1e-3
    ye = gemv_abi(m, xt, EXACT)   # same rounded weights as the generic kernel
    assert ((ye.float() - g.float()).abs().max() / ref.abs().max()).item() <= 5e-4
    # W sub-block bit-exact vs oracle
    Wb = tensor_to_bits(W).reshape(L.num_indices, 8, H)[rows].reshape(48 * 8, H)
    assert (Wb == vo.dequant(sub)).all()


def test_grouped_launch_matches_single(dev):
    from vptq_amd import _backend as B
    import ctypes as C
    shapes = [(2048, 1024), (2048, 256), (2048, 256)]       # q / k / v of one decoder layer
    mods, xs, singles = [], [], []
    x = bits_to_tensor(vo.from_f32(np.random.default_rng(0).standard_normal((1, 1, 2048))
                                   .astype(np.float32), "f16"), "f16", dev).reshape(1, 1, 2048)
    for i, (I, O) in enumerate(shapes):
        m = spec_to_module(vo.make_layer(I, O, dist="llm", seed=20 + i), dev)
        mods.append(m)
        singles.append(m(x))
    descs = (B.LayerDesc * len(mods))()
    keep = []
    for i, m in enumerate(mods):
        from _gpu_util import module_desc
        d, k = module_desc(m)
        descs[i] = d
        keep.append(k)
    ys = [torch.empty_like(s) for s in singles]
    xp = (C.c_void_p * 3)(*[x.data_ptr()] * 3)
    yp = (C.c_void_p * 3)(*[t.data_ptr() for t in ys])
    B.check(B.lib().vptq_quant_gemv_grouped(descs, 3, xp, yp, 1, module_flags(),
                                            B.current_stream_ptr(dev)), "grouped")
    torch.cuda.synchronize()
    for a, b in zip(singles, ys):
        assert torch.equal(a, b)


def test_grouped_launch_mixed_shapes_and_tokens(dev):
    """One grouped launch over layers of different widths (the sweep count is chosen by the
    widest), different heights, 2 tokens, one of them with an odd row count."""
    from vptq_amd import _backend as B
    from _gpu_util import module_desc
    import ctypes as C
    shapes = [(8192 + 1024, 264), (1024, 2048), (4096, 512), (2048, 40)]
    mods, xs, want = [], [], []
    for i, (I, O) in enumerate(shapes):
        L = vo.make_layer(I, O, dist="llm", seed=60 + i, bias=(i % 2 == 0))
        m = spec_to_module(L, dev)
        xb = vo.from_f32(np.random.default_rng(i).standard_normal((1, 2, I)).astype(np.float32), "f16")
        mods.append(m)
        xs.append(bits_to_tensor(xb, "f16", dev).reshape(1, 2, I))
        want.append(vo.forward(L, xb))
    descs = (B.LayerDesc * len(mods))()
    keep = []
    for i, m in enumerate(mods):
        d, k = module_desc(m)
        descs[i] = d
        keep.append(k)
    ys = [torch.empty(1, 2, O, device=dev, dtype=torch.float16) for _, O in shapes]
    xp = (C.c_void_p * 4)(*[t.data_ptr() for t in xs])
    yp = (C.c_void_p * 4)(*[t.data_ptr() for t in ys])
    B.check(B.lib().vptq_quant_gemv_grouped(descs, 4, xp, yp, 2, 0, B.current_stream_ptr(dev)),
            "grouped")
    torch.cuda.synchronize()
    for y, w in zip(ys, want):
        assert rel_err(tensor_to_bits(y), w, "f16") <= 1e-3


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("shapes,tokens", [
    ([(2048, 4096), (2048, 1024), (2048, 1024)], 2),      # 192 row groups, one sweep: 2 token slots on the matrix pipe
    ([(2048, 4096), (2048, 1024)], 4),                    # 4 slots (one sweep)
    ([(2048, 8192), (2048, 8192), (2048, 8192)], 1),      # three equal layers of 256 row groups: issued as {A, B} + {C}
    ([(1024, 8192), (1024, 8192), (1024, 2048), (1024, 2048)], 1),
])
def test_grouped_launch_reference_roundings(shapes, tokens, dt, dev):
    """vptq_quant_gemv_grouped with VPTQ_GEMV_EXACT over sibling-shaped groups (round 6: several tokens in the reference's roundings
    on the matrix pipe, groups split into the cheapest set of launches): every layer against the oracle and bit for bit against
    its own single launch of the same kernel family."""
    from vptq_amd import _backend as B
    from _gpu_util import module_desc
    import ctypes as C
    n = len(shapes)
    I = shapes[0][0]
    xb = vo.from_f32(np.random.default_rng(n + tokens).standard_normal((1, tokens, I)).astype(np.float32), dt)
    x = bits_to_tensor(xb, dt, dev).reshape(1, tokens, I)
    mods, want = [], []
    for i, (_, O) in enumerate(shapes):
        L = vo.make_layer(I, O, dist="llm", seed=90 + i, dtype=dt, bias=(i == 1))
        mods.append(spec_to_module(L, dev))
        want.append(vo.forward(L, xb))
    descs = (B.LayerDesc * n)()
    keep = []
    for i, m in enumerate(mods):
        d, k = module_desc(m)
        descs[i] = d
        keep.append(k)
    assert B.lib().vptq_quant_gemv_grouped_kernel_name(descs, n, tokens, EXACT) == b"gemv_k256m_kernel"
    ys = [torch.empty(1, tokens, O, device=dev, dtype=x.dtype) for _, O in shapes]
    xp = (C.c_void_p * n)(*[x.data_ptr()] * n)
    yp = (C.c_void_p * n)(*[t.data_ptr() for t in ys])
    B.check(B.lib().vptq_quant_gemv_grouped(descs, n, xp, yp, tokens, EXACT, B.current_stream_ptr(dev)), "grouped")
    torch.cuda.synchronize()
    for m, y, w in zip(mods, ys, want):
        assert rel_err(tensor_to_bits(y), w, dt) <= (5e-4 if dt == "f16" else TOL[dt])
        assert bit_identical_frac(tensor_to_bits(y), w) >= 0.9
        assert torch.equal(y, gemv_abi(m, x, EXACT | MFMA))   # the same loop, whatever the launch's share of the CUs


def test_read_ahead_hint_does_not_change_results(dev):
    """chain_prefetch is a pure performance hint (also with a range smaller / larger than
    the layer's own index tensor and with perm)."""
    import vptq_amd
    mods = [spec_to_module(vo.make_layer(2048, O, dist="llm", seed=40 + i, enable_perm=(i == 1)), dev)
            for i, O in enumerate((1024, 512, 2048))]
    x = torch.randn(1, 2, 2048, device=dev, dtype=torch.float16)
    plain = [m(x) for m in mods]
    vptq_amd.layers.chain_prefetch(mods, circular=True)
    assert mods[0]._prefetch_next is mods[1] and mods[2]._prefetch_next is mods[0]
    hinted = [m(x) for m in mods]
    torch.cuda.synchronize()
    for a, b in zip(plain, hinted):
        assert torch.equal(a, b)
    assert "_prefetch_next" not in mods[0].state_dict() and len(list(mods[0].children())) <= 3


def test_absorb_perm_on_device(dev):
    from vptq_amd.utils.pack import absorb_perm_layer
    L = vo.make_layer(2048, 512, dist="llm", seed=77, enable_perm=True)
    m = spec_to_module(L, dev)
    x = torch.randn(1, 1, 2048, device=dev, dtype=torch.float16,
                    generator=torch.Generator(device=dev).manual_seed(5))
    W0, y0 = m.dequant(), m(x)
    assert absorb_perm_layer(m)
    assert torch.equal(m.dequant(), W0)                    # identical dense weight
    y1 = m(x)   # same weights, columns summed in another order: one flipped fp16 bit at most
    assert ((y1.float() - y0.float()).abs().max() / y0.float().abs().max()).item() <= 1e-3


def test_deterministic_and_stream_safe(dev):
    """Fixed summation order => bit-identical results run to run, also when two streams
    launch different layers concurrently (the library keeps no per-call state)."""
    mods = [spec_to_module(vo.make_layer(4096, 1024, dist="llm", seed=90 + i), dev) for i in range(2)]
    x = torch.randn(1, 1, 4096, device=dev, dtype=torch.float16)
    ref = [m(x).clone() for m in mods]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for _ in range(25):
        outs = []
        for m, st in zip(mods, streams):
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                outs.append(m(x))
        torch.cuda.synchronize()
        for o, r in zip(outs, ref):
            assert torch.equal(o, r)


def test_hipgraph_capture_of_the_forward(dev):
    L = vo.make_layer(4096, 4096, dist="llm", seed=1)
    m = spec_to_module(L, dev)
    x = torch.randn(1, 1, 4096, device=dev, dtype=torch.float16)
    eager = m(x)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        m(x)  # warm-up on the side stream
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            out = m(x)
    x.copy_(torch.randn_like(x))
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, m(x)) and not torch.equal(out, eager)


def test_loader_end_to_end_vs_dense_model(dev, tmp_path):
    """vptq.AutoModelForCausalLM.from_pretrained on a synthetic VPTQ checkpoint: logits of
    the quantised model == logits of the same model with every VQuantLinear replaced by a
    dense nn.Linear holding its dequantised weight (decode and prefill routes)."""
    import copy
    import vptq_amd
    from _ckpt import write_tiny_checkpoint
    write_tiny_checkpoint(str(tmp_path))
    model = vptq_amd.AutoModelForCausalLM.from_pretrained(str(tmp_path), device=str(dev),
                                                          link_prefetch=True)
    dense = copy.deepcopy(model)
    for name, m in list(dense.named_modules()):
        if isinstance(m, vptq_amd.VQuantLinear):
            lin = torch.nn.Linear(m.in_features, m.out_features, bias=False, device=dev,
                                  dtype=torch.float16)
            lin.weight.data = m.dequant()
            dense.set_submodule(name, lin)
    with torch.no_grad():
        for T in (1, 3, 12):
            ids = torch.randint(0, model.config.vocab_size, (1, T), device=dev)
            a = model(input_ids=ids).logits.float()
            b = dense(input_ids=ids).logits.float()
            assert torch.isfinite(a).all()
            assert ((a - b).abs().max() / b.abs().max()).item() <= 5e-3, T


# ---------------------------------------------------------------- v2 wire format
V2_CONFIGS = [
    dict(in_features=1024, out_features=2048, num_centroids=8192, num_res_centroids=256),  # uint8 ids
    dict(in_features=1024, out_features=1024, num_centroids=8192, num_res_centroids=512),  # uint16 ids
    # >= 1024 vector-rows: one token takes the MFMA variant of the LDS-resident kernel (folded form)
    dict(in_features=1024, out_features=8192, num_centroids=8192, num_res_centroids=256),
    dict(in_features=2048 + 64, out_features=8192 + 8 * 3, num_centroids=4096, num_res_centroids=512),
]


@pytest.mark.parametrize("cfg", V2_CONFIGS)
@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_quant_gemv_v2_reference_test_shapes(cfg, dtype, dev):
    """Same data recipe as the reference's only kernel test
    (tests/test_quant_gemv.py:112-171): seed 1234, normal(0.02, 0.5), cyclic ids."""
    import vptq_amd
    torch.manual_seed(1234)
    I, O, k, kr, v = (cfg["in_features"], cfg["out_features"], cfg["num_centroids"],
                      cfg["num_res_centroids"], 8)
    dt = TORCH_DT[dtype]

    def nrm(*shape):
        return torch.normal(mean=0.02, std=0.5, size=shape, device=dev, dtype=dt)

    x, cent, rcent = nrm(1, 1, I), nrm(1, k, v), nrm(1, kr, v)
    sw, sb = nrm(I, 1), nrm(I, 1)
    n = I * O // v
    ids = torch.arange(k, device=dev, dtype=torch.int32).repeat(n // k + 1)[:n].to(torch.uint16)
    rids = torch.arange(kr, device=dev, dtype=torch.int32).repeat(n // kr + 1)[:n].to(
        torch.uint16 if kr > 256 else torch.uint8)
    out = vptq_amd.ops.quant_gemv_v2(
        x=x, bias=None, indices=ids, centroids=cent, residual_indices=rids,
        residual_centroids=rcent, scale_weights=sw, scale_bias=sb, vector_len=v,
        num_codebooks=1, num_centroids=k, num_residual_centroids=kr, out_features=O)
    want = vo.gemv_v2_ground_truth(
        tensor_to_bits(x), None, ids.cpu().numpy().astype(np.int64), tensor_to_bits(cent),
        rids.cpu().numpy().astype(np.int64), tensor_to_bits(rcent), tensor_to_bits(sw),
        tensor_to_bits(sb), v, O, dtype)
    got = tensor_to_bits(out)
    assert rel_err(got, want, dtype) <= TOL[dtype]
    # the reference's own (loose) criterion: rtol = atol = 0.2
    a, b = vo.to_f32(got, dtype), vo.to_f32(want, dtype)
    assert np.allclose(a, b, rtol=0.2, atol=0.2)


def test_sibling_groups_share_one_launch(dev):
    """link_siblings: q/k/v (and gate/up) of a block are computed by the first one called; the
    others hand out their share for the SAME tensor object and launch on their own otherwise."""
    import vptq_amd

    class Block(torch.nn.Module):
        def __init__(self):
            super().__init__()
            mk = lambda O, seed: spec_to_module(vo.make_layer(1024, O, dist="llm", seed=seed), dev)  # noqa: E731
            self.q_proj, self.k_proj, self.v_proj = mk(1024, 1), mk(256, 2), mk(256, 3)
            self.gate_proj, self.up_proj = mk(2048, 4), mk(2048, 5)
            self.o_proj = mk(1024, 6)

    blk = Block()
    names = ["q_proj", "k_proj", "v_proj", "gate_proj", "up_proj", "o_proj"]
    x = bits_to_tensor(vo.from_f32(np.random.default_rng(0).standard_normal((1, 1, 1024))
                                   .astype(np.float32), "f16"), "f16", dev).reshape(1, 1, 1024)
    x2 = x * 0.5
    want = {n: getattr(blk, n)(x) for n in names}
    want2 = {n: getattr(blk, n)(x2) for n in names}
    assert vptq_amd.layers.link_siblings(blk) == 2
    assert blk.q_proj._siblings is blk.v_proj._siblings and "_siblings" not in blk.o_proj.__dict__
    calls = []
    lib = vptq_amd._backend.lib()
    real = lib.vptq_quant_gemv_grouped
    grp = blk.q_proj._siblings

    def same(a, b):   # 1 fp16 ulp of the largest output: 3+ tokens use the exact arithmetic
        return rel_err(tensor_to_bits(a), tensor_to_bits(b), "f16") <= 1e-3

    # usual order; then another order; then the same member twice; then a different tensor between
    for order, xs in ((names, [x] * 6), (["v_proj", "q_proj", "k_proj", "up_proj", "gate_proj"], [x] * 5),
                      (["q_proj", "q_proj", "k_proj"], [x] * 3),
                      (["q_proj", "k_proj", "v_proj"], [x, x2, x])):
        for n, xi in zip(order, xs):
            got = getattr(blk, n)(xi)
            assert same(got, (want if xi is x else want2)[n]), n
    # an in-place update of x invalidates what the group still holds
    xm = x.clone()
    q = blk.q_proj(xm)
    xm.mul_(0.5)
    k = blk.k_proj(xm)
    assert same(q, want["q_proj"]) and same(k, want2["k_proj"])
    # 2 tokens through the group, 8 tokens past it
    xt = torch.cat([x, x2], dim=1)
    qq, kk = blk.q_proj(xt), blk.k_proj(xt)
    assert same(qq[:, :1], want["q_proj"]) and same(kk[:, 1:], want2["k_proj"])
    x8 = x.expand(1, 8, 1024).contiguous()
    assert same(blk.v_proj(x8)[:, 3:4], want["v_proj"])
    # under hipGraph capture the members of a group become one kernel node
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            outs = [getattr(blk, n)(x) for n in names]
        g.replay()
    torch.cuda.synchronize()
    for n, o in zip(names, outs):
        assert same(o, want[n]), n
    assert grp._x is None or grp._out == {}


# ---------------------------------------------------------------- BASELINE sizes vs the reference
from _cases import big_names, load_big, v2_names, load_v2, fmt_names, load_fmt, stored_rows  # noqa: E402


@pytest.mark.parametrize("name", big_names())
def test_baseline_size_goldens_every_kernel_and_form(name, dev):
    """hidden 4096 / 8192 layers against the REAL reference's outputs (procedural inputs,
    tests/golden/gen_golden_big.py): dense W bit for bit (sha256), forward through both k = 256
    kernels in both arithmetic forms, the generic kernel and the module's own route."""
    L, x, y, cfg, W_head = load_big(name)
    if cfg["tokens"] > 4:
        pytest.skip("many tokens: test_many_token_routes_vs_reference_goldens")
    dt = cfg["dtype"]
    m = spec_to_module(L, dev)
    W = tensor_to_bits(m.dequant())
    assert (W[:2] == W_head).all()
    assert hashlib.sha256(W.tobytes()).hexdigest() == cfg["W_sha256"]
    del W
    xt = bits_to_tensor(x, dt, dev).reshape(x.shape)
    out = tensor_to_bits(m(xt))
    assert rel_err(out, y, dt) <= TOL[dt], f"module route: {rel_err(out, y, dt):.3e}"
    seen = {}
    for flags in (0, EXACT, MFMA, VALU, MFMA | EXACT, VALU | EXACT, GENERIC):
        got = tensor_to_bits(gemv_abi(m, xt, flags))
        err = rel_err(got, y, dt)
        seen[flags] = (kernel_name(m, cfg["tokens"], flags), err, bit_identical_frac(got, y))
        assert err <= TOL[dt], (flags, seen[flags])
    # the reference's roundings reproduce (nearly) its bits; fp32-accumulated order aside
    if dt == "f16":
        assert seen[EXACT][2] >= 0.95 and seen[GENERIC][2] >= 0.95, seen
    # (bf16, round 6: its roundings on the matrix pipe / as dot blocks - the same bits as the reference's widened arithmetic)
    assert seen[EXACT][2] >= 0.95 and seen[MFMA | EXACT][2] >= 0.95 and seen[VALU | EXACT][2] >= 0.95, seen
    print(name, seen)


@pytest.mark.parametrize("name", [n for n in big_names() if n.endswith(("_t64", "_t256", "_t8192"))])
def test_many_token_routes_vs_reference_goldens(name, dev):
    """64, 256 and 8192 (BASELINE configs[3]: batch 4 x seq 2048, bf16) tokens through a 4096^2 layer against the REAL reference's prefill branch (dequant +
    F.linear, /root/reference/vptq/ops/quant_gemm.py:231-274; fixtures: tests/golden/gen_golden_big.py):
    the module's own route, the cached dense route, the fused dequant-tile GEMM, and - 64 tokens, fp16 -
    the batched-decode kernel (4 launches of 16 tokens)."""
    from vptq_amd import ops
    L, x, y, cfg, W_head = load_big(name)
    dt, T = cfg["dtype"], cfg["tokens"]
    m = spec_to_module(L, dev)
    assert hashlib.sha256(tensor_to_bits(m.dequant()).tobytes()).hexdigest() == cfg["W_sha256"]
    xt = bits_to_tensor(x, dt, dev).reshape(x.shape)
    routes = {"module": lambda: m(xt), "dense_cached": lambda: m._dense_cached(xt),
              "gemm_fused": lambda: ops.quant_gemm_fused(xt, m._descriptor()[1], L.out_features)}
    if T <= 64:
        # the batched-decode kernels (launches of 16 tokens): the one-pass kernel (default from 5 tokens, both dtypes) and
        # the kernel with the reference's roundings
        assert kernel_name(m, T) == "gemm_k256t_kernel"
        routes["gemm_k256t"] = lambda: gemv_abi(m, xt, 0)
        assert kernel_name(m, T, EXACT) == "gemm_k256_kernel"      # (bf16 since round 6)
        routes["gemm_k256"] = lambda: gemv_abi(m, xt, EXACT)
    errs = {}
    for rname, fn in routes.items():
        out = stored_rows(tensor_to_bits(fn()).reshape(1, T, -1), cfg)
        errs[rname] = rel_err(out, y, dt)
        assert errs[rname] <= TOL[dt], (rname, errs)
    print(name, {k: f"{v:.2e}" for k, v in errs.items()})


FMT_KERNELS = {"t1_k8192_r256_8192x1024": "gemv_lds_mfma_kernel", "t1_k4096_r512_perm_bias": "gemv_lds_mfma_kernel",
               "t1_k8192_r256_bf16": "gemv_lds_mfma_kernel", "t8_k65536_r256": "gemv_gather_kernel",
               "t5_k65536_r65536_perm": "gemv_gather_kernel", "t7_v6_k4096_r16": "gemv_gatherx_kernel",
               "t1_v10_k4096_r256": "gemv_gatherx_kernel", "t2_v2_k256_r16_perm": "gemv_gatherx_kernel",
               "t8_v4_k4096_r256_bf16": "gemv_gatherx_kernel",
               # round 3: one-token fixtures of the k = 65536 formats (also served by gemv_sliced over the derived layout:
               # tests/test_gemv_sliced_gpu.py)
               "t1_k65536_r0_4096x4096": "gemv_gather_kernel", "t1_k65536_r0_bf16": "gemv_gather_kernel",
               "t1_k65536_r256_4096x4096": "gemv_gather_kernel", "t1_k65536_r256_bf16": "gemv_gather_kernel",
               # round 4: the two-table ("4 bit") format and vector length 16 (sliced: two passes / 32-byte entries)
               "t1_k65536_r65536_4096x4096": "gemv_gather_kernel", "t1_k65536_r65536_bf16_perm": "gemv_gather_kernel",
               "t1_v16_k65536_r65536_4096x4096": "gemv_gatherx_kernel", "t1_v16_k65536_r0_bf16_perm": "gemv_gatherx_kernel",
               "t1_v16_k65536_r1024_4096x4096": "gemv_gatherx_kernel", "t1_v8_k65536_r4_bias": "gemv_gatherx_kernel",
               "t1_v8_k32768_r0_perm": "gemv_gatherx_kernel",
               # 2 - 4 tokens of large-codebook layers (the module's route: one launch over the sliced layouts where that was
               # measured faster, tests/test_gemv_sliced_gpu.py; through vptq_quant_gemv: the gather kernels)
               "t2_k65536_r256_4096x4096": "gemv_gather_kernel", "t4_k65536_r0_8192x2048_perm": "gemv_gather_kernel",
               "t3_k65536_r65536_bf16": "gemv_gather_kernel", "t2_v16_k65536_r65536_4096x2048": "gemv_gatherx_kernel"}


@pytest.mark.parametrize("name", fmt_names())
def test_reference_goldens_other_formats(name, dev):
    """The kernels of the non-canonical formats against the REAL reference's outputs (procedural inputs,
    tests/golden/gen_golden_fmt.py): dense W bit for bit (sha256); forward through the library's kernel
    choice (the one named in FMT_KERNELS), with the reference's roundings, through the generic kernel and
    through the module's own route."""
    L, x, y, cfg, W_head = load_fmt(name)
    dt, tokens = cfg["dtype"], cfg["tokens"]
    m = spec_to_module(L, dev)
    W = tensor_to_bits(m.dequant())
    assert (W[:2] == W_head).all()
    assert hashlib.sha256(W.tobytes()).hexdigest() == cfg["W_sha256"]
    del W
    assert kernel_name(m, tokens) == FMT_KERNELS[name], kernel_name(m, tokens)
    xt = bits_to_tensor(x, dt, dev).reshape(x.shape)
    out = tensor_to_bits(m(xt))
    assert rel_err(out, y, dt) <= TOL[dt], f"module route: {rel_err(out, y, dt):.3e}"
    seen = {}
    for flags in (0, EXACT, GENERIC):
        got = tensor_to_bits(gemv_abi(m, xt, flags))
        err = rel_err(got, y, dt)
        seen[flags] = (kernel_name(m, tokens, flags), err, bit_identical_frac(got, y))
        assert err <= TOL[dt], (flags, seen[flags])
    if dt == "f16":
        assert seen[EXACT][2] >= 0.95 and seen[GENERIC][2] >= 0.95, seen
    print(name, seen)


# ---------------------------------------------------------------- v2: the reference test's data
@pytest.mark.parametrize("name", v2_names())
def test_quant_gemv_v2_reference_fixture(name, dev):
    """The reference's kernel test (tests/test_quant_gemv.py:222-247) replayed: ITS data
    (create_test_data, seed 1234), ITS call `vptq.ops.quant_gemv_v2(**test_data)` through the
    `vptq` alias package, ITS expected output (`ground_truth`, stored by
    tests/golden/gen_golden_v2.py) and ITS comparison (rtol = atol = 0.2) - plus ours."""
    import vptq
    d = load_v2(name)
    c = d["cfg"]
    dt = c["dtype"]
    tdt = TORCH_DT[dt]
    I, O, k, kr, v, T = (c["in_features"], c["out_features"], c["num_centroids"],
                         c["num_res_centroids"], c["vector_len"], c["length"])
    test_data = {
        "x": bits_to_tensor(d["x"], dt, dev).reshape(1, T, I),
        "bias": None if d["bias"] is None else bits_to_tensor(d["bias"], dt, dev).reshape(1, O),
        "indices": torch.from_numpy(d["indices"].view(np.int16).copy()).to(dev).view(torch.uint16),
        "centroids": bits_to_tensor(d["centroids"], dt, dev).reshape(1, k, v),
        "residual_indices": (torch.from_numpy(d["res_indices"].view(np.int16).copy()).to(dev).view(torch.uint16)
                             if d["res_indices"].dtype == np.uint16 else
                             torch.from_numpy(d["res_indices"].copy()).to(dev)),
        "residual_centroids": bits_to_tensor(d["res_centroids"], dt, dev).reshape(1, kr, v),
        "scale_weights": bits_to_tensor(d["scale_weights"], dt, dev).reshape(I, 1),
        "scale_bias": bits_to_tensor(d["scale_bias"], dt, dev).reshape(I, 1),
        "vector_len": v, "num_codebooks": 1, "num_centroids": k, "num_residual_centroids": kr,
        "out_features": O,
    }
    out2 = vptq.ops.quant_gemv_v2(**test_data)
    assert out2.dtype == tdt and tuple(out2.shape) == (1, T, O)
    # 15 tokens (the op's maximum) = chunks of 4 (fp16) / 2 (bf16) through the same kernel
    x15 = test_data["x"][:, :1].expand(1, 15, I).contiguous()
    out15 = vptq.ops.quant_gemv_v2(**dict(test_data, x=x15))
    assert torch.equal(out15[:, 14], out2[:, 0]) and torch.equal(out15[:, 3], out2[:, 0])
    got = tensor_to_bits(out2)
    a, b = vo.to_f32(got, dt), vo.to_f32(d["y"], dt)
    assert np.allclose(a, b, rtol=0.2, atol=0.2)          # the reference's criterion
    assert rel_err(got, d["y"], dt) <= TOL[dt]            # ours


# ---------------------------------------------------------------- derived state must not go stale
def test_in_place_parameter_updates_invalidate_the_descriptor(dev):
    """forward -> in-place update of scale / bias / perm (load_state_dict's copy_, an optimizer
    step) -> forward: the cached descriptor holds pointers to DERIVED copies (scale / bias in
    column order for perm layers) and must be rebuilt (ADVICE medium)."""
    L = vo.make_layer(1024, 512, dist="llm", seed=5, enable_perm=True, bias=True)
    m = spec_to_module(L, dev)
    x = bits_to_tensor(vo.from_f32(np.random.default_rng(3).standard_normal((1, 1, 1024))
                                   .astype(np.float32), "f16"), "f16", dev).reshape(1, 1, 1024)
    xb = tensor_to_bits(x)
    # (the many-token route - cached dequant descriptor with argsort(perm) + F.linear - follows too)
    x70 = bits_to_tensor(vo.from_f32(np.random.default_rng(8).standard_normal((1, 70, 1024))
                                     .astype(np.float32), "f16"), "f16", dev).reshape(1, 70, 1024)
    x70b = tensor_to_bits(x70)
    assert rel_err(tensor_to_bits(m(x)), vo.forward(L, xb), "f16") <= 1e-3
    assert rel_err(tensor_to_bits(m(x70)), vo.forward(L, x70b), "f16") <= 1e-3
    # 1. in-place scaling of weight_scale
    with torch.no_grad():
        m.weight_scale.mul_(2.0)
    assert rel_err(tensor_to_bits(m(x)), vo.forward(module_to_spec(m), xb), "f16") <= 1e-3
    assert rel_err(tensor_to_bits(m(x70)), vo.forward(module_to_spec(m), x70b), "f16") <= 1e-3
    # 2. load_state_dict (default assign=False -> copy_ into the same storages)
    L2 = vo.make_layer(1024, 512, dist="llm", seed=6, enable_perm=True, bias=True)
    m2 = spec_to_module(L2, dev)
    ptr = m.weight_scale.data_ptr()
    m.load_state_dict(m2.state_dict())
    assert m.weight_scale.data_ptr() == ptr
    assert rel_err(tensor_to_bits(m(x)), vo.forward(L2, xb), "f16") <= 1e-3
    # 3. a new permutation written in place
    with torch.no_grad():
        m.perm.copy_(torch.from_numpy(np.random.default_rng(9).permutation(1024).astype(np.uint16)
                                      .view(np.int16)).to(dev))
    assert rel_err(tensor_to_bits(m(x)), vo.forward(module_to_spec(m), xb), "f16") <= 1e-3
    assert rel_err(tensor_to_bits(m(x70)), vo.forward(module_to_spec(m), x70b), "f16") <= 1e-3
    # sibling groups copy descriptors by value: they follow as well
    import vptq_amd

    class Blk(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.gate_proj, self.up_proj = m, m2

    blk = Blk()
    assert vptq_amd.layers.link_siblings(blk) == 1
    xs = x.clone()
    a0 = blk.gate_proj(xs); b0 = blk.up_proj(xs)
    with torch.no_grad():
        m2.weight_bias.add_(0.25)
    xs2 = x.clone()
    blk.gate_proj(xs2); b1 = blk.up_proj(xs2)
    assert rel_err(tensor_to_bits(b1), vo.forward(module_to_spec(m2), xb), "f16") <= 1e-3
    assert not torch.equal(b0, b1)


def test_forward_under_inference_mode(dev):
    """Tensors made under torch.inference_mode() track no version counter (reading `_version`
    raises): perm layers, sibling groups and models LOADED under inference mode must work."""
    import vptq_amd
    with torch.inference_mode():
        L = vo.make_layer(1024, 512, dist="llm", seed=7, enable_perm=True)
        m = spec_to_module(L, dev)
        L2 = vo.make_layer(1024, 256, dist="llm", seed=8, enable_perm=True)
        m2 = spec_to_module(L2, dev)

        class Blk(torch.nn.Module):
            def __init__(self):
                super().__init__()
                self.gate_proj, self.up_proj = m, m2

        blk = Blk()
        assert vptq_amd.layers.link_siblings(blk) == 1
        x = bits_to_tensor(vo.from_f32(np.random.default_rng(4).standard_normal((1, 1, 1024))
                                       .astype(np.float32), "f16"), "f16", dev).reshape(1, 1, 1024)
        assert x.is_inference()
        ya, yb = blk.gate_proj(x), blk.up_proj(x)
        y17 = m(x.expand(1, 17, 1024).contiguous())     # dequant + GEMM route (inverse perm cache)
    xb = tensor_to_bits(x)
    assert rel_err(tensor_to_bits(ya), vo.forward(L, xb), "f16") <= 1e-3
    assert rel_err(tensor_to_bits(yb), vo.forward(L2, xb), "f16") <= 1e-3
    assert rel_err(tensor_to_bits(y17[:, 5:6]), vo.forward(L, xb), "f16") <= 1e-3


# ---------------------------------------------------------------- tensor-parallel shards on the GPU
@pytest.mark.parametrize("world", [2, 4, 8])
def test_shards_through_the_hip_kernels(world, dev):
    """All `world` shards of shard_out_features / shard_in_features run through the HIP kernels on
    one GPU: concatenated slices == the full layer bit for bit; fp32 partial sums
    (VPTQ_GEMV_OUT_F32) added up and rounded once == the full layer to 1e-3 and the oracle."""
    from vptq_amd.utils.shard import shard_in_features, shard_out_features, forward_partial_f32
    for (I, O, bias) in ((4096, 4096, True), (8192, 1024, False)):
        L = vo.make_layer(I, O, dist="llm", seed=31 + world, bias=bias)
        m = spec_to_module(L, dev)
        for T in (1, 3):
            x = bits_to_tensor(vo.from_f32(np.random.default_rng(T).standard_normal((1, T, I))
                                           .astype(np.float32), "f16"), "f16", dev).reshape(1, T, I)
            full = m(x)
            # column parallel (disjoint output slices): exact (same kernel family per slice aside,
            # so compare against the oracle-tolerance as well as for equality where it holds)
            cols = torch.cat([shard_out_features(m, r, world)(x) for r in range(world)], dim=-1)
            assert rel_err(tensor_to_bits(cols), tensor_to_bits(full), "f16") <= 1e-3
            # row parallel: fp32 partials, one rounding
            acc = torch.zeros(1, T, O, dtype=torch.float32, device=dev)
            for r in range(world):
                s = shard_in_features(m, r, world)
                assert (s.bias is not None) == (bias and r == 0)
                part = forward_partial_f32(s, x[..., s.shard[1]:s.shard[2]].contiguous())
                assert part.dtype == torch.float32
                acc += part
            rows = acc.to(torch.float16)
            assert rel_err(tensor_to_bits(rows), tensor_to_bits(full), "f16") <= 1e-3
            want = vo.forward(L, tensor_to_bits(x))
            assert rel_err(tensor_to_bits(rows), want, "f16") <= 1e-3
    # fp32 output of the unsharded layer == its fp16 output before rounding, every kernel family
    for kw in (dict(), dict(num_centroids=65536, num_res_centroids=0), dict(vector_len=4, num_res_centroids=0)):
        L = vo.make_layer(512, 256, dist="llm", seed=3, **kw)
        m = spec_to_module(L, dev)
        x = bits_to_tensor(vo.from_f32(np.random.default_rng(0).standard_normal((1, 2, 512))
                                       .astype(np.float32), "f16"), "f16", dev).reshape(1, 2, 512)
        p = forward_partial_f32(m, x)
        assert torch.equal(p.to(torch.float16), m(x))


@pytest.mark.parametrize("world", [1, 2, 8])
def test_row_parallel_sibling_groups_in_one_launch(world, dev):
    """bench.py --mode tp_row: the row-parallel shards of sibling projections (q / k / v: one input,
    different row counts) go out as ONE grouped launch with fp32 partial outputs into one contiguous
    buffer (vptq_quant_gemv_grouped + VPTQ_GEMV_OUT_F32), are summed over the ranks (the all-reduce) and
    rounded once: every projection must come out as the full layer does, and as the oracle says."""
    import ctypes as C
    from vptq_amd import _backend as B
    from vptq_amd.utils.shard import shard_in_features
    I, Os = 4096, (2048, 256, 256)
    Ls = [vo.make_layer(I, O, dist="llm", seed=50 + i) for i, O in enumerate(Os)]
    ms = [spec_to_module(L, dev) for L in Ls]
    xb = vo.from_f32(np.random.default_rng(5).standard_normal((1, 1, I)).astype(np.float32), "f16")
    x = bits_to_tensor(xb, "f16", dev).reshape(1, 1, I)
    lib = B.lib()
    acc = torch.zeros(1, 1, sum(Os), dtype=torch.float32, device=dev)
    for r in range(world):
        shards = [shard_in_features(m, r, world) for m in ms]
        xs = x[..., shards[0].shard[1]:shards[0].shard[2]].contiguous()
        part = torch.full((1, 1, sum(Os)), float("nan"), dtype=torch.float32, device=dev)
        ds, keep = [], []
        for s in shards:
            d, k = module_desc(s)
            ds.append(d); keep.append(k)
        offs = [sum(Os[:j]) for j in range(len(Os))]
        arr = (B.LayerDesc * 3)(*ds)
        xp = (C.c_void_p * 3)(*[xs.data_ptr()] * 3)
        yp = (C.c_void_p * 3)(*[part.data_ptr() + 4 * o for o in offs])
        rc = lib.vptq_quant_gemv_grouped(arr, 3, xp, yp, 1, B.GEMV_OUT_F32, B.current_stream_ptr(dev))
        assert rc == 0, lib.vptq_last_error()
        torch.cuda.synchronize()
        assert torch.isfinite(part).all()
        acc += part
    y = acc.to(torch.float16)
    o = 0
    for L, m, O in zip(Ls, ms, Os):
        got = tensor_to_bits(y[..., o:o + O].contiguous())
        assert rel_err(got, tensor_to_bits(m(x)), "f16") <= 1e-3
        assert rel_err(got, vo.forward(L, xb), "f16") <= 1e-3
        o += O


# ---------------------------------------------------------------- LDS-resident codebooks (k <= 8192)
LDS_CASES = [
    # I, O, kwargs, tokens     (T = index_bits + res_bits: 12, 13, 20, 21 (two splits), 22)
    (1024, 512, dict(num_centroids=4096, num_res_centroids=0), 1),
    (1024, 520, dict(num_centroids=8192, num_res_centroids=0, bias=True), 2),
    (2048, 1024, dict(num_centroids=4096, num_res_centroids=256, dist="llm"), 1),
    (520, 136, dict(num_centroids=4096, num_res_centroids=512, enable_perm=True), 3),     # T=21, ib=12
    (1024, 2048, dict(num_centroids=8192, num_res_centroids=256), 1),                      # reference test shape
    (360, 72, dict(num_centroids=8192, num_res_centroids=256, enable_perm=True, bias=True), 4),
    (1024, 1024, dict(num_centroids=8192, num_res_centroids=512, dist="llm"), 1),          # T=22
    (4096 + 8, 264, dict(num_centroids=8192, num_res_centroids=512), 5),                   # ragged chunk, 5 tokens
    (512, 40, dict(num_centroids=1024, num_res_centroids=4, enable_norm=False), 1),        # T=12 as 10 + 2
    (8192, 8192, dict(num_centroids=8192, num_res_centroids=256, dist="llm"), 1),          # BASELINE size
    (1024, 4096 * 8, dict(num_centroids=4096, num_res_centroids=256, dist="llm"), 2),      # 16 rows per group
    (1024, 512, dict(num_centroids=8192, num_res_centroids=256, dtype="bf16", dist="llm"), 1),
    (776, 200, dict(num_centroids=4096, num_res_centroids=512, dtype="bf16", dist="llm", enable_perm=True, bias=True), 3),
]


@pytest.mark.parametrize("I,O,kw,tokens", LDS_CASES)
def test_lds_resident_kernel_vs_oracle(I, O, kw, tokens, dev):
    """v = 8, one codebook, 256 < k <= 8192, kr <= 512: both codebooks LDS-resident
    (gemv_lds.hip; the reference's v2 kernel design, csrc/kernels/quant_gemv_v2.cuh:85-94, for the
    packed format).  fp16 = the reference's roundings (>= 95 % of the outputs bit-identical)."""
    kw = dict(kw)
    dist = kw.pop("dist", "ref-test")
    L = vo.make_layer(I, O, dist=dist, seed=I + O + 7, **kw)
    dt = L.dtype
    rng = np.random.default_rng(11)
    xs = (0.02 + 0.5 * rng.standard_normal((1, tokens, I))) if dist == "ref-test" \
        else rng.standard_normal((1, tokens, I))
    x = vo.from_f32(xs.astype(np.float32), dt)
    m = spec_to_module(L, dev)
    # one token on >= 1024 vector-rows: the MFMA variant (folded form); else the reference's roundings
    folded = tokens == 1 and O >= 8192 and I <= 8192
    assert kernel_name(m, tokens) == ("gemv_lds_mfma_kernel" if folded else "gemv_lds_kernel"), kernel_name(m, tokens)
    assert kernel_name(m, tokens, GENERIC) == "gemv_generic_kernel"
    if dt == "bf16":   # the reference's roundings for bf16: the L2-gather kernel has them
        assert kernel_name(m, tokens, EXACT) == "gemv_gatherx_kernel"
    else:
        assert kernel_name(m, tokens, EXACT) == "gemv_lds_kernel"
    xt = bits_to_tensor(x, dt, dev).reshape(x.shape)
    if I * O <= 4096 * 4096:
        W_ref = vo.dequant(L, ref_residual_mask_quirk=False)
        want = vo.gemv(W_ref, x, dt, L.bias)
    else:
        from oracle import c_oracle as co
        want = co.forward(L, x, quirk=False)
    got = tensor_to_bits(m(xt))
    assert got.shape == want.shape
    err = rel_err(got, want, dt)
    assert err <= TOL[dt], f"{err:.3e}"
    if dt == "f16":
        ex = tensor_to_bits(gemv_abi(m, xt, EXACT))
        assert bit_identical_frac(ex, want) >= 0.95
        if not folded or (module_flags() & EXACT):
            assert bit_identical_frac(got, want) >= 0.95
    # same as the generic kernel (the A/B partner) and as fp32 output rounded once
    gen = tensor_to_bits(gemv_abi(m, xt, GENERIC))
    assert rel_err(got, gen, dt) <= TOL[dt]
    from vptq_amd.utils.shard import forward_partial_f32
    assert torch.equal(forward_partial_f32(m, xt).to(xt.dtype), m(xt))


LDS_MFMA_CASES = [
    # I, O, kwargs: one token, >= 1024 vector-rows -> gemv_lds_mfma_kernel (folded form, MFMA accumulate)
    (1024, 8192, dict(num_centroids=4096, num_res_centroids=0)),                                 # T = 12, no residual
    (1000, 8200, dict(num_centroids=8192, num_res_centroids=0, bias=True)),                      # T = 13, ragged step, spare rows
    (2048, 8192, dict(num_centroids=4096, num_res_centroids=256, dist="llm", enable_perm=True)),  # T = 20, permutation
    (520, 8192 + 16, dict(num_centroids=4096, num_res_centroids=512, enable_perm=True, bias=True)),  # T = 21 as 12 + 9
    (4096, 8192, dict(num_centroids=8192, num_res_centroids=512, dist="llm")),                   # T = 22
    (512, 16384, dict(num_centroids=1024, num_res_centroids=4, enable_norm=False)),              # no norm, 8 rows per group
    (256, 8 * 4096 + 24, dict(num_centroids=4096, num_res_centroids=256, dist="llm", bias=True)),  # 16 rows per group, spare rows
    (8192 + 2048, 8192, dict(num_centroids=4096, num_res_centroids=256, dist="llm")),            # two staging passes
    (1024, 8192, dict(num_centroids=8192, num_res_centroids=256, dtype="bf16", dist="llm")),
    (776, 8192 + 8, dict(num_centroids=4096, num_res_centroids=512, dtype="bf16", dist="llm", enable_perm=True, bias=True)),
]


@pytest.mark.parametrize("I,O,kw", LDS_MFMA_CASES)
def test_lds_resident_mfma_kernel_vs_oracle(I, O, kw, dev):
    """gemv_lds_mfma_kernel: one token in the default (folded) arithmetic with the multiply-accumulate on
    the matrix pipe, codebooks LDS-resident, f16(s x) staged; against the oracle (<= 1e-3 / 8e-3), the
    kernel with the reference's roundings (VPTQ_GEMV_EXACT), the generic kernel; deterministic; fp32 output."""
    kw = dict(kw)
    dist = kw.pop("dist", "ref-test")
    L = vo.make_layer(I, O, dist=dist, seed=I + O + 5, **kw)
    dt = L.dtype
    rng = np.random.default_rng(23)
    xs = (0.02 + 0.5 * rng.standard_normal((1, 1, I))) if dist == "ref-test" else rng.standard_normal((1, 1, I))
    x = vo.from_f32(xs.astype(np.float32), dt)
    m = spec_to_module(L, dev)
    assert kernel_name(m, 1) == "gemv_lds_mfma_kernel", kernel_name(m, 1)
    assert kernel_name(m, 2) == "gemv_lds_kernel"
    xt = bits_to_tensor(x, dt, dev).reshape(x.shape)
    from oracle import c_oracle as co
    want = co.forward(L, x, quirk=False)
    got = tensor_to_bits(gemv_abi(m, xt, 0))
    err = rel_err(got, want, dt)
    assert err <= TOL[dt], f"{err:.3e}"
    gen = tensor_to_bits(gemv_abi(m, xt, GENERIC))
    assert rel_err(got, gen, dt) <= TOL[dt]
    if dt == "f16":
        ex = tensor_to_bits(gemv_abi(m, xt, EXACT))
        assert bit_identical_frac(ex, want) >= 0.95 and rel_err(got, ex, dt) <= TOL[dt]
    outs = [tensor_to_bits(gemv_abi(m, xt, 0)) for _ in range(3)]
    assert all((o == got).all() for o in outs)
    from vptq_amd import _backend as B
    y32 = torch.empty(1, 1, O, dtype=torch.float32, device=dev)
    desc, keep = module_desc(m)
    B.check(B.lib().vptq_quant_gemv(desc, xt.data_ptr(), y32.data_ptr(), 1, B.GEMV_OUT_F32, None, 0,
                                    B.current_stream_ptr(dev)), "vptq_quant_gemv")
    assert (tensor_to_bits(y32.to(xt.dtype)) == got).all()


def test_lds_kernel_golden_and_determinism(dev):
    L, x, y, cfg, _ = load_golden("k8192_r256_t21")
    m = spec_to_module(L, dev)
    assert kernel_name(m, cfg["tokens"]) == "gemv_lds_kernel"
    xt = bits_to_tensor(x, cfg["dtype"], dev).reshape(x.shape)
    outs = [tensor_to_bits(m(xt)) for _ in range(10)]
    assert all((o == outs[0]).all() for o in outs)
    assert rel_err(outs[0], y, cfg["dtype"]) <= TOL[cfg["dtype"]]
    assert bit_identical_frac(outs[0], y) >= 0.9


# ---------------------------------------------------------------- L2-gather kernel for the other large formats
GATHERX_CASES = [
    # I, O, kwargs, tokens      v = 8 / 16, any index width T, several codebook groups, no outliers
    (1024, 512, dict(vector_len=16, num_centroids=65536, num_res_centroids=65536, dist="llm"), 1),   # v16-k65536-65536, T = 32
    (1024, 520, dict(vector_len=16, num_centroids=65536, num_res_centroids=32768, dist="llm", bias=True), 2),  # T = 31, padded O
    (2048, 256, dict(vector_len=16, num_centroids=65536, num_res_centroids=1024, dist="llm"), 1),    # T = 26
    (1024, 256, dict(vector_len=16, num_centroids=65536, num_res_centroids=0), 3),                   # T = 16
    (1028, 264, dict(vector_len=8, num_centroids=32768, num_res_centroids=0, enable_perm=True), 1),  # T = 15, ragged piece
    (520, 136, dict(vector_len=8, num_centroids=65536, num_res_centroids=1024, enable_perm=True, bias=True), 4),  # T = 26
    (1024, 512, dict(vector_len=8, num_centroids=16384, num_res_centroids=16384, dist="llm"), 1),    # T = 28
    (2048, 128, dict(vector_len=8, num_centroids=4096, num_res_centroids=4096, num_codebooks=2), 2),  # two codebook groups, T = 24
    (1024, 64, dict(vector_len=16, num_centroids=256, num_res_centroids=256, num_codebooks=4, enable_perm=True), 1),
    (512, 48, dict(vector_len=8, num_centroids=1024, num_res_centroids=4, enable_norm=False, num_codebooks=2), 5),  # no norm, 5 tokens
    (4, 8, dict(vector_len=8, num_centroids=256, num_res_centroids=0), 1),                             # one piece of one lane
    # outlier columns with a codebook of the same vector length
    (1024 + 128, 512, dict(vector_len=8, num_centroids=65536, num_res_centroids=256, outlier_size=128, outlier_vector_len=8,
                           num_outlier_centroids=1024, dist="llm"), 1),
    (512 + 64, 264, dict(vector_len=16, num_centroids=65536, num_res_centroids=4096, outlier_size=64, outlier_vector_len=16,
                         num_outlier_centroids=256, enable_perm=True, bias=True), 3),
    (1024 + 300, 96, dict(vector_len=12, num_centroids=4096, num_res_centroids=0, outlier_size=300, outlier_vector_len=12,
                          num_outlier_centroids=4096, enable_perm=True, dtype="bf16", dist="llm"), 2),   # 300 > 256 outlier columns
    (512 + 32, 128, dict(vector_len=8, num_centroids=1024, num_res_centroids=1024, num_codebooks=2, outlier_size=32,
                         outlier_vector_len=8, num_outlier_centroids=16, enable_norm=False), 1),
    (1024, 768, dict(vector_len=12, num_centroids=65536, num_res_centroids=4096, dist="llm"), 1),     # v12-k65536-4096, T = 28
    (1032, 100, dict(vector_len=12, num_centroids=4096, num_res_centroids=0, enable_perm=True, bias=True), 4),  # v = 12, padded O
    (512, 120, dict(vector_len=12, num_centroids=65536, num_res_centroids=256, dtype="bf16", dist="llm"), 2),
    (8192, 8192, dict(vector_len=16, num_centroids=65536, num_res_centroids=65536, dist="llm"), 1),  # BASELINE size
    (1024, 512, dict(vector_len=16, num_centroids=65536, num_res_centroids=4096, dtype="bf16", dist="llm"), 1),
    (776, 200, dict(vector_len=8, num_centroids=32768, num_res_centroids=512, dtype="bf16", dist="llm", enable_perm=True, bias=True), 3),
    # the other vector lengths the reference dispatches (csrc/quant_gemv.cu:210-233): 2, 4, 6, 10
    (1024, 384, dict(vector_len=6, num_centroids=4096, num_res_centroids=0, dist="llm"), 1),          # v6-k4096-0, T = 12
    (1032, 100, dict(vector_len=6, num_centroids=65536, num_res_centroids=256, enable_perm=True, bias=True), 3),  # padded O, T = 24
    (512, 120, dict(vector_len=6, num_centroids=1024, num_res_centroids=1024, num_codebooks=2, dtype="bf16", dist="llm"), 2),
    (2048, 256, dict(vector_len=4, num_centroids=65536, num_res_centroids=0, dist="llm"), 1),          # 8-byte entries
    (520, 132, dict(vector_len=4, num_centroids=256, num_res_centroids=16, enable_perm=True, bias=True), 4),
    (1024, 64, dict(vector_len=4, num_centroids=4096, num_res_centroids=4096, dtype="bf16", enable_norm=False), 2),
    (1024, 250, dict(vector_len=10, num_centroids=65536, num_res_centroids=1024, dist="llm", bias=True), 1),   # 20-byte entries, T = 26
    (516, 90, dict(vector_len=10, num_centroids=4096, num_res_centroids=0, enable_perm=True, dtype="bf16"), 3),
    (1024, 66, dict(vector_len=2, num_centroids=256, num_res_centroids=0), 1),                           # 4-byte entries
    (512, 34, dict(vector_len=2, num_centroids=4096, num_res_centroids=64, enable_perm=True, bias=True, dist="llm"), 2),
    (512 + 32, 96, dict(vector_len=6, num_centroids=4096, num_res_centroids=0, outlier_size=32, outlier_vector_len=6,
                        num_outlier_centroids=64), 1),
    # outlier codebooks of vector length 4 - the only one the reference's kernel takes (csrc/quant_gemv.cu:186-189)
    (1024 + 128, 512, dict(vector_len=8, num_centroids=65536, num_res_centroids=256, outlier_size=128, outlier_vector_len=4,
                           num_outlier_centroids=4096, dist="llm"), 1),
    (512 + 64, 260, dict(vector_len=8, num_centroids=4096, num_res_centroids=4096, outlier_size=64, outlier_vector_len=4,
                         num_outlier_centroids=256, enable_perm=True, bias=True), 3),        # padded O: outlier rows end inside a vector-row
    (1024 + 300, 96, dict(vector_len=12, num_centroids=4096, num_res_centroids=0, outlier_size=300, outlier_vector_len=4,
                          num_outlier_centroids=1024, enable_perm=True, dtype="bf16", dist="llm"), 2),
    (512 + 32, 136, dict(vector_len=16, num_centroids=65536, num_res_centroids=1024, outlier_size=32, outlier_vector_len=4,
                         num_outlier_centroids=16, enable_norm=False), 4),                   # O = 136: 8.5 vector-rows of 16
    # one token of a long vector on few vector-rows and >= 4096 columns
    (4096 + 8, 1200, dict(vector_len=12, num_centroids=65536, num_res_centroids=4096, dist="llm", enable_perm=True, bias=True), 1),
    (4096, 1000, dict(vector_len=10, num_centroids=4096, num_res_centroids=256, dtype="bf16", dist="llm"), 1),
    # 5-8 tokens in one launch for v <= 8 (8 token slots), two launches of 4 for the longer vectors
    (1024, 264, dict(vector_len=8, num_centroids=32768, num_res_centroids=512, dist="llm", bias=True), 7),
    (520, 132, dict(vector_len=4, num_centroids=65536, num_res_centroids=0, enable_perm=True), 8),
    (1032, 96, dict(vector_len=6, num_centroids=4096, num_res_centroids=4096, num_codebooks=2, dtype="bf16", dist="llm"), 6),
    (1024, 256, dict(vector_len=16, num_centroids=65536, num_res_centroids=65536, dist="llm"), 6),
    (512 + 64, 128, dict(vector_len=8, num_centroids=4096, num_res_centroids=4096, outlier_size=64, outlier_vector_len=4,
                         num_outlier_centroids=256, enable_perm=True), 5),
]


@pytest.mark.parametrize("I,O,kw,tokens", GATHERX_CASES)
def test_gatherx_kernel_vs_oracle(I, O, kw, tokens, dev):
    """Every vector length the reference dispatches (2 ... 16), any codebook sizes (total index width
    8 ... 32), one or several codebook groups, outlier columns with a codebook of the same vector length
    or of vector length 4: gemv_gatherx.hip (L2 gathers, window unpack for any width) instead of the
    generic kernel - the reference's template space csrc/quant_gemv.cu:42-132, 186-233.  Reference roundings."""
    kw = dict(kw)
    dist = kw.pop("dist", "ref-test")
    L = vo.make_layer(I, O, dist=dist, seed=I + O + 13, **kw)
    dt = L.dtype
    rng = np.random.default_rng(17)
    xs = (0.02 + 0.5 * rng.standard_normal((1, tokens, I))) if dist == "ref-test" \
        else rng.standard_normal((1, tokens, I))
    x = vo.from_f32(xs.astype(np.float32), dt)
    m = spec_to_module(L, dev)
    m.enable_sliced_layout(False)   # (this test is the gather kernel's: v16-k65536-0 / -65536 would take the sliced layout)
    assert kernel_name(m, tokens) == "gemv_gatherx_kernel", kernel_name(m, tokens)
    assert kernel_name(m, tokens, GENERIC) == "gemv_generic_kernel"
    xt = bits_to_tensor(x, dt, dev).reshape(x.shape)
    if I * O <= 4096 * 4096:
        W_ref = vo.dequant(L, ref_residual_mask_quirk=False)
        want = vo.gemv(W_ref, x, dt, L.bias)
    else:
        from oracle import c_oracle as co
        want = co.forward(L, x, quirk=False)
    got = tensor_to_bits(m(xt))
    assert got.shape == want.shape
    err = rel_err(got, want, dt)
    assert err <= TOL[dt], f"{err:.3e}"
    if dt == "f16":
        assert bit_identical_frac(got, want) >= 0.95
    gen = tensor_to_bits(gemv_abi(m, xt, GENERIC))
    assert rel_err(got, gen, dt) <= TOL[dt]
    from vptq_amd.utils.shard import forward_partial_f32
    assert torch.equal(forward_partial_f32(m, xt).to(xt.dtype), m(xt))
    outs = [tensor_to_bits(m(xt)) for _ in range(3)]
    assert all((o == outs[0]).all() for o in outs)


# ---------------------------------------------------------------- batched decode: 5-16 tokens, one launch
GEMM_CASES = [
    # I, O, kwargs, tokens
    (1024, 256, dict(bias=True), 5),
    (2048, 1032, dict(), 16),                       # rows not a multiple of the 4-row group
    (4104, 264, dict(enable_perm=True), 9),         # ragged last tile of 1024 columns + permutation
    (8192, 8192, dict(dist="llm"), 16),             # BASELINE size, one row group per workgroup
    (1024, 12288, dict(dist="llm", bias=True), 7),  # 384 row groups: 2 per workgroup on some CUs
    (512, 8 * 4 * 1100, dict(dist="llm"), 13),      # 1100 row groups: two passes of <= 4 groups
    (14336, 512, dict(dist="llm"), 6),              # 14 tiles
]


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("I,O,kw,tokens", GEMM_CASES)
def test_batched_decode_kernel_vs_oracle(I, O, kw, tokens, dt, dev):
    """gemm_k256_kernel (canonical format, 5-16 tokens in one launch, VPTQ_GEMV_EXACT: since round 3 the default
    for these token counts is the one-pass gemm_k256t, tests/test_gemm_k256t_gpu.py): dequantised tiles with
    the reference's roundings -> LDS -> v_mfma_f32_16x16x16_f16 / _bf16, i.e. the arithmetic of the
    reference's own path for these token counts, dequant + F.linear (quant_gemm.py:231-274).  bf16 since round 6 (its
    roundings as blocks of v_dot2_f32_bf16)."""
    kw = dict(kw)
    dist = kw.pop("dist", "ref-test")
    L = vo.make_layer(I, O, dist=dist, seed=I + O + tokens, dtype=dt, **kw)
    rng = np.random.default_rng(tokens)
    xs = (0.02 + 0.5 * rng.standard_normal((1, tokens, I))) if dist == "ref-test" \
        else rng.standard_normal((1, tokens, I))
    x = vo.from_f32(xs.astype(np.float32), dt)
    m = spec_to_module(L, dev)
    tdt = torch.float16 if dt == "f16" else torch.bfloat16
    assert kernel_name(m, tokens, EXACT) == "gemm_k256_kernel" and kernel_name(m, tokens) == "gemm_k256t_kernel"
    xt = bits_to_tensor(x, dt, dev).reshape(x.shape)
    got = gemv_abi(m, xt, EXACT)
    assert rel_err(tensor_to_bits(m(xt)), tensor_to_bits(got), dt) <= TOL[dt]   # (the module's default route: the one-pass kernel)
    # bit-level partner on the GPU: HIP dequant (bit-exact W) + fp32 matmul, rounded once
    W = m.dequant().float()
    ref = (xt.float() @ W.t())
    if m.bias is not None:
        ref = ref + m.bias.float()
    ref16 = tensor_to_bits(ref.to(tdt))
    gb = tensor_to_bits(got)
    assert rel_err(gb, ref16, dt) <= TOL[dt] and bit_identical_frac(gb, ref16) >= 0.95   # (fp16: 1 ulp of the largest output = 9.8e-4)
    if I * O <= 2048 * 2048:
        assert rel_err(gb, vo.forward(L, x), dt) <= TOL[dt]
    # every token row equals that token alone through the one-token kernel (exact form)
    one = tensor_to_bits(gemv_abi(m, xt[:, tokens - 1:tokens].contiguous(), EXACT))
    assert rel_err(gb[:, tokens - 1:tokens], one, dt) <= TOL[dt]
    # fp32 output
    assert torch.equal(gemv_abi(m, xt, EXACT, out_f32=True).to(tdt), got)
    from vptq_amd.utils.shard import forward_partial_f32
    assert torch.equal(forward_partial_f32(m, xt).to(tdt), m(xt))   # (default route, fp32 partial sums)
    # determinism
    assert torch.equal(gemv_abi(m, xt, EXACT), got)


def test_batched_decode_golden_16_tokens(dev):
    L, x, y, cfg, _ = load_golden("canon_t16_perm")
    m = spec_to_module(L, dev)
    assert kernel_name(m, 16, EXACT) == "gemm_k256_kernel" and kernel_name(m, 4).startswith("gemv_k256")
    xt = bits_to_tensor(x, "f16", dev).reshape(x.shape)
    out = tensor_to_bits(gemv_abi(m, xt, EXACT))
    assert rel_err(out, y, "f16") <= 1e-3 and bit_identical_frac(out, y) >= 0.9
    assert kernel_name(m, 16) == "gemm_k256t_kernel"   # the default: one pass, folded arithmetic, permutation applied by its pre-pass
    assert rel_err(tensor_to_bits(m(xt)), y, "f16") <= 1e-3


# ---------------------------------------------------------------- fused dequant + GEMM (prefill)
FUSED_CASES = [
    # I, O, kwargs, tokens
    (1024, 1024, dict(), 128),
    (1024, 1000, dict(bias=True), 300),            # ragged N tile (125 vector-rows), ragged M tile
    (4104, 264, dict(dist="llm"), 77),             # K not a multiple of the 64-column step
    (2048, 512, dict(dtype="bf16", dist="llm", bias=True), 200),
    (8192, 2048, dict(dist="llm"), 512),
    (512, 8192, dict(dtype="bf16", dist="llm"), 1000),
]


@pytest.mark.parametrize("I,O,kw,tokens", FUSED_CASES)
def test_fused_dequant_gemm_vs_dequant_plus_matmul(I, O, kw, tokens, dev):
    """vptq_quant_gemm (dequantised tile -> LDS -> 32x32x16 MFMA) against the reference's structure for
    many tokens, dequant + matmul (quant_gemm.py:231-274): HIP dequant (bit-exact W) + fp32 matmul."""
    from vptq_amd import _backend as B
    from vptq_amd import ops
    kw = dict(kw)
    dist = kw.pop("dist", "ref-test")
    L = vo.make_layer(I, O, dist=dist, seed=I + O + tokens, **kw)
    dt = L.dtype
    rng = np.random.default_rng(tokens)
    xs = (0.02 + 0.5 * rng.standard_normal((1, tokens, I))) if dist == "ref-test" \
        else rng.standard_normal((1, tokens, I))
    x = vo.from_f32(xs.astype(np.float32), dt)
    m = spec_to_module(L, dev)
    desc, keep = module_desc(m)
    assert B.lib().vptq_quant_gemm_supported(desc) == 1
    xt = bits_to_tensor(x, dt, dev).reshape(x.shape)
    got = ops.quant_gemm_fused(xt, desc, O)
    W = m.dequant().float()
    ref = xt.float() @ W.t()
    if m.bias is not None:
        ref = ref + m.bias.float()
    gb, rb = tensor_to_bits(got), tensor_to_bits(ref.to(xt.dtype))
    err = rel_err(gb, rb, dt)
    assert err <= TOL[dt], f"{err:.3e}"
    if dt == "f16":
        assert bit_identical_frac(gb, rb) >= 0.95
    # the module's forward takes this path when it is enabled (VPTQ_FUSED_GEMM_MAX_TOKENS; off by
    # default: hipBLASLt wins on MI355X) and the dense route otherwise
    import sys
    qg = sys.modules["vptq_amd.ops.quant_gemm"]     # (the package attribute of that name is the function)
    old = qg._FUSED_GEMM_MAX_TOKENS
    try:
        qg._FUSED_GEMM_MAX_TOKENS = 4096
        assert torch.equal(m(xt), got)
        qg._FUSED_GEMM_MAX_TOKENS = 0
        assert rel_err(tensor_to_bits(m(xt)), rb, dt) <= TOL[dt]
    finally:
        qg._FUSED_GEMM_MAX_TOKENS = old
    # a few rows against the CPU oracle
    if I * O <= 2048 * 1024:
        want = vo.forward(L, x[:, :3])
        assert rel_err(gb[:, :3], want, dt) <= TOL[dt]


def test_fused_gemm_unsupported_layers_fall_back(dev):
    from vptq_amd import _backend as B
    L = vo.make_layer(512, 256, dist="llm", seed=3, enable_perm=True)
    m = spec_to_module(L, dev)
    desc, keep = module_desc(m)
    assert B.lib().vptq_quant_gemm_supported(desc) == 0
    x = torch.randn(1, 100, 512, device=dev, dtype=torch.float16)
    y = m(x)                                    # dequant + F.linear
    ref = x.float() @ m.dequant().float().t()
    assert rel_err(tensor_to_bits(y), tensor_to_bits(ref.half()), "f16") <= 1e-3
    L2 = vo.make_layer(512, 256, dist="llm", seed=3, num_centroids=4096, num_res_centroids=0)
    d2, k2 = module_desc(spec_to_module(L2, dev))
    assert B.lib().vptq_quant_gemm_supported(d2) == 0
    rc = B.lib().vptq_quant_gemm(d2, x.data_ptr(), y.data_ptr(), 100, 0, None, 0, None)
    assert rc == -3


# ---------------------------------------------------------------- Hugging Face's own loading route
def test_hf_from_pretrained_vptq_route(dev, tmp_path):
    """transformers.AutoModelForCausalLM.from_pretrained on a synthetic VPTQ checkpoint
    (config.json with quantization_config + model.safetensors): HF's VptqHfQuantizer builds
    `vptq.VQuantLinear` (= this package through the `vptq` alias) on meta and loads the tensors by
    state-dict name; logits equal those of this package's own loader and of the dense model.
    (reference flow: vptq/layers/model_base.py:93-199; HF: integrations/vptq.py, quantizers/quantizer_vptq.py)"""
    import vptq
    import vptq_amd
    from _ckpt import write_tiny_checkpoint
    from _hf_route import hf_from_pretrained
    write_tiny_checkpoint(str(tmp_path), perm=True)
    model, upstream_bug = hf_from_pretrained(str(tmp_path), dtype=torch.float16, device_map={"": dev.index or 0})
    print("transformers' replace_with_vptq_linear needed the one-line fix:", upstream_bug)
    model = model.eval()
    qlayers = [m for m in model.modules() if isinstance(m, vptq_amd.VQuantLinear)]
    assert len(qlayers) == 2 * 7 and all(type(m) is vptq.VQuantLinear for m in qlayers)
    assert all(m.indices.is_cuda and m.indices.dtype == torch.int32 and m.perm.dtype == torch.int16 for m in qlayers)
    assert isinstance(model.lm_head, torch.nn.Linear)
    own = vptq_amd.AutoModelForCausalLM.from_pretrained(str(tmp_path), device=str(dev))
    with torch.no_grad():
        for T in (1, 3, 7, 20):
            ids = torch.randint(0, model.config.vocab_size, (1, T), device=dev)
            a, b = model(input_ids=ids).logits.float(), own(input_ids=ids).logits.float()
            assert torch.isfinite(a).all()
            assert ((a - b).abs().max() / b.abs().max()).item() <= 2e-3, T
        # decode loop through HF generate (greedy): same tokens from both models
        ids = torch.randint(0, model.config.vocab_size, (1, 5), device=dev)
        ga = model.generate(ids, max_new_tokens=8, do_sample=False)
        gb = own.generate(ids, max_new_tokens=8, do_sample=False)
        assert ga.shape == (1, 13) and (ga == gb).float().mean().item() >= 0.9


# ---------------------------------------------------------------- `python -m vptq`
def test_command_line_prompt_and_chat(dev, tmp_path, capsys):
    """The reference's command line (vptq/app_utils.py:56-110, 165-189) on a synthetic checkpoint with a
    word-level tokenizer: a prompt completion (100 greedy tokens, streamed) equals `generate` on this
    package's loader, and a scripted two-turn chat goes through the tokenizer's chat template."""
    import vptq_amd
    import vptq_amd.app_utils as app
    from _ckpt import write_tiny_checkpoint, write_tiny_tokenizer
    write_tiny_checkpoint(str(tmp_path), perm=False)
    tok = write_tiny_tokenizer(str(tmp_path))
    out = app.main(["--model", str(tmp_path), "--prompt", "w1 w2 w3 w4"])
    text = capsys.readouterr().out
    assert out.shape == (1, 4 + 100) and "w1 w2 w3 w4" in text
    model = vptq_amd.AutoModelForCausalLM.from_pretrained(str(tmp_path), device=str(dev))
    ids = tok("w1 w2 w3 w4", return_tensors="pt").input_ids.to(dev)
    want = model.generate(ids, max_new_tokens=100, do_sample=False, pad_token_id=2)
    assert (out[:, :want.shape[1]].cpu() == want.cpu()).float().mean().item() >= 0.9
    assert any(isinstance(m, vptq_amd.VQuantLinear) for m in model.modules())
    # chat: scripted input, two turns, then `exit`
    lines = iter(["w10 w11", "w12", "exit"])
    args = app.define_basic_args().parse_args(["--model", str(tmp_path), "--chat", "--chat-system-prompt", "w9"])
    torch.manual_seed(0)
    history = app.chat_loop(model, tok, args, read=lambda prompt: next(lines))
    roles = [m["role"] for m in history]
    assert roles == ["system", "user", "assistant", "user", "assistant"], roles
    assert history[0]["content"] == "w9" and history[3]["content"] == "w12"
    assert "Press 'exit' to quit" in capsys.readouterr().out
    # a tokenizer without a chat template falls back to the prompt completion
    tok.chat_template = None
    out2 = app.chat_loop(model, tok, args)
    assert out2.shape[1] > 1 and "no chat_template" in capsys.readouterr().out


# ---------------------------------------------------------------- adversarial families for the default arithmetic
def _adversarial_layer(I, O, family, seed):
    """canonical fp16 layers built to stress the folded arithmetic:
    "plain"; "cyclic" = the reference test's index pattern (/root/reference/tests/test_quant_gemv.py:21-31);
    "bias4" / "bias16" = |weight_bias| = 4x / 16x rms(weight_scale * (centroid + residual));
    "bias0.7" = the same construction below the load-time gate; "llm" = tensors shaped like a checkpoint's (bias well below
    the scaled weights) - since round 4 the reference test's own distribution (every tensor normal(0.02, 0.5): bias ratio
    1.4) is above the gate's line of 1, so "plain" is served in the reference's roundings"""
    L = vo.make_layer(I, O, dist="llm" if family == "llm" else "ref-test", seed=seed)
    rng = np.random.default_rng(seed + 1)
    if family == "cyclic":
        N, G = L.num_indices, L.group_size
        idx = (np.arange(N * G) % 256).reshape(1, N, G)
        L.indices = vo.pack_indices(idx, 8, idx.copy(), 8)
    if family.startswith("bias"):
        ratio = float(family[4:])
        c, r, s = (vo.to_f32(t, "f16") for t in (L.centroids, L.res_centroids, L.weight_scale))
        ws = np.sqrt((c ** 2).mean() + (r ** 2).mean()) * np.sqrt((s ** 2).mean())
        L.weight_bias = vo.from_f32((rng.standard_normal(I) * ratio * ws).astype(np.float32), "f16")
    return L


def _adversarial_x(L, kind, seed):
    I = L.in_features
    rng = np.random.default_rng(seed)
    if kind == "large_mean":
        xs = 2.0 + 0.5 * rng.standard_normal(I)
    else:
        xs = 0.02 + 0.5 * rng.standard_normal(I)
    if kind == "orthogonal":   # sum b x cancels: what is left of y is the part the reference computes worst
        b = vo.to_f32(L.weight_bias, "f16").astype(np.float64)
        xs = xs - b * (xs @ b) / (b @ b)
    return vo.from_f32(xs.astype(np.float32).reshape(1, 1, I), "f16")


ADVERSARIAL = [("plain", "normal"), ("plain", "large_mean"), ("llm", "normal"), ("llm", "large_mean"), ("cyclic", "normal"),
               ("cyclic", "large_mean"), ("bias0.7", "orthogonal"), ("bias4", "normal"), ("bias4", "orthogonal"),
               ("bias16", "orthogonal")]


@pytest.mark.parametrize("family,xkind", ADVERSARIAL)
@pytest.mark.parametrize("O", [512, 4608])   # the VALU kernel / the persistent MFMA kernel (144 row groups)
def test_adversarial_families_reference_default(family, xkind, O, dev):
    """The product default (the reference's roundings for every layer) on inputs built against the folded form:
    inside the bar by a wide margin, module forward = the C ABI with VPTQ_GEMV_EXACT bit for bit."""
    import vptq_amd
    assert vptq_amd.arithmetic() == "reference"
    L = _adversarial_layer(4096, O, family, seed=77 + O)
    x = _adversarial_x(L, xkind, seed=5)
    want = vo.forward(L, x)
    m = spec_to_module(L, dev)
    xt = bits_to_tensor(x, "f16", dev).reshape(x.shape)
    assert m._descriptor()[9] == EXACT
    y = m(xt)
    assert torch.equal(y.view(torch.int16), gemv_abi(m, xt, EXACT).view(torch.int16))
    assert rel_err(tensor_to_bits(y), want, "f16") <= 1e-3      # (at most the flip of one last bit: 2^-10 of max|y|)
    assert bit_identical_frac(tensor_to_bits(y), want) >= 0.95


@pytest.mark.parametrize("family,xkind", ADVERSARIAL)
@pytest.mark.parametrize("O", [512, 4608])   # the VALU kernel / the persistent MFMA kernel (144 row groups)
def test_adversarial_families_default_route(family, xkind, O, dev, folded_arithmetic):
    """The OPT-IN folded arithmetic through the module must stay inside the 1e-3 bar on inputs built against it:
    bias-dominated layers are recognised at load time (VQuantLinear._folded_form_is_safe: both forms on probe
    activations) and served in the reference's arithmetic; everything else takes the folded form and must pass as it is."""
    from vptq_amd.ops.chain import GemvChain
    I = 4096
    L = _adversarial_layer(I, O, family, seed=77 + O)
    x = _adversarial_x(L, xkind, seed=5)
    want = vo.forward(L, x)
    m = spec_to_module(L, dev)
    xt = bits_to_tensor(x, "f16", dev).reshape(x.shape)
    gated = m._descriptor()[9] != 0
    # (round 4: the cyclic pattern with in_features a multiple of k makes every index row the same - 8 distinct
    # outputs; layers with fewer than 32 distinct vector-rows are gated as well: tools/gpu_fuzz_count.py)
    # round 5: the gate MEASURES the folded form against the reference's roundings on probe activations (float32 outputs):
    # bias-dominated layers and the cyclic pattern sit far above its line, LLM-like tensors and bias0.7 far below; the
    # reference test's own distribution ("plain": bias as large as the scaled weights) sits near it and may go either way
    if family != "plain":
        assert gated == (family in ("bias4", "bias16", "cyclic")), (family, gated)
    err = rel_err(tensor_to_bits(m(xt)), want, "f16")
    assert err <= 1e-3, f"module default route, {family}/{xkind}: {err:.2e}"
    # the chain API with the same layers (the gate routes bias-dominated ones to the per-layer exact path)
    ys = GemvChain([m, m])([xt, xt], flags=None if gated else 8)
    err_c = rel_err(tensor_to_bits(ys[1]), want, "f16")
    assert err_c <= 1e-3, f"chain route, {family}/{xkind}: {err_c:.2e}"
    # the folded form itself through the C ABI, whatever the gate says (reported; asserted only below the gate)
    raw = rel_err(tensor_to_bits(gemv_abi(m, xt, 0)), want, "f16")
    exact = rel_err(tensor_to_bits(gemv_abi(m, xt, EXACT)), want, "f16")
    print(f"\n[adversarial] O={O} {family}/{xkind}: default route {err:.2e} (gated={gated}), chain {err_c:.2e}, "
          f"folded form {raw:.2e}, reference roundings {exact:.2e}")
    assert exact <= 5e-4
    if not gated:
        assert raw <= 1e-3
    # the functional op (vptq.ops.quant_gemm) applies the same gate, and several tokens (the one-pass batched kernel
    # is folded arithmetic too) go through it as well
    import importlib
    qg = importlib.import_module("vptq_amd.ops.quant_gemm")   # (the package also exports a function of that name)
    assert (qg._safe_flags(m.indices, m.centroids.weight, m.res_centroids.weight, m.weight_scale, m.weight_bias,
                           m._descriptor()[1], I, O) != 0) == gated
    x6 = np.concatenate([x] * 6, axis=1)
    x6t = bits_to_tensor(x6, "f16", dev).reshape(x6.shape)
    y6 = m(x6t)
    assert rel_err(tensor_to_bits(y6)[:, 5:6], want, "f16") <= 1e-3, f"6 tokens, {family}/{xkind}"
    assert kernel_name(m, 6, m._descriptor()[9]) == ("gemm_k256_kernel" if gated else "gemm_k256t_kernel")
