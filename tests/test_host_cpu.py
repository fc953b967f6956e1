"""CPU-side checks: C ABI exports, host format tools, module contract, loud failure."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import vptq_amd
from vptq_amd import _backend as B
from vptq_amd.utils.pack import pack_index, unpack_index_tensor
from oracle import vptq_oracle as vo
from _cases import golden_names, load_golden
from _gpu_util import spec_to_module
from _refshim import reference_available

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "vptq_hip.h")).read()
    return sorted(set(re.findall(r"VPTQ_API[^;(]*?\b(vptq_\w+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    syms = _declared_symbols()
    assert len(syms) >= 8, syms
    assert os.path.exists(B.LIB_PATH), "build libvptq_hip.so first (__graft_entry__.build())"
    lib = ctypes.CDLL(B.LIB_PATH)
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/vptq_hip.h but not exported"
    assert sorted(B.EXPORTS) == syms, "python binding table out of sync with the header"
    lib.vptq_abi_version.restype = ctypes.c_int
    assert lib.vptq_abi_version() == B.ABI_VERSION == 10


def test_ctypes_struct_layout_matches_header():
    # 16 int32 + 13 pointers + int64; 8 int32 + 7 pointers
    assert ctypes.sizeof(B.LayerDesc) == 16 * 4 + 13 * 8 + 8
    assert ctypes.sizeof(B.V2Desc) == 8 * 4 + 7 * 8
    hdr = open(os.path.join(ROOT, "include", "vptq_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    body = hdr[hdr.index("typedef struct VptqLayerDesc {"):hdr.index("} VptqLayerDesc;")]
    names = re.findall(r"\b(\w+);", body)
    assert names == [n for n, _ in B.LayerDesc._fields_]
    body = hdr[hdr.index("typedef struct VptqV2Desc {"):hdr.index("} VptqV2Desc;")]
    names = re.findall(r"\b(\w+);", body)
    assert names == [n for n, _ in B.V2Desc._fields_]


def _canonical_desc(I=4096, O=4096, dtype=0, perm=False):
    """A descriptor of the canonical 2-bit format with fake (aligned, never dereferenced) pointers."""
    d = B.LayerDesc()
    d.in_features, d.out_features, d.vector_len, d.num_codebooks, d.group_size = I, O, 8, 1, I
    d.num_centroids, d.num_res_centroids, d.index_bits, d.res_bits = 256, 256, 8, 8
    d.row_words, d.num_indices, d.dtype = I // 2, O // 8, dtype
    d.indices, d.centroids, d.res_centroids = 1 << 20, 2 << 20, 3 << 20
    d.weight_scale, d.weight_bias = 4 << 20, 5 << 20
    if perm:
        d.perm, d.scale_permuted, d.bias_permuted = 6 << 20, 7 << 20, 8 << 20
    return d


def test_chain_routing_without_gpu():
    """vptq_quant_gemv_chain_kernel_name is host logic: one persistent launch when every layer is of the
    canonical format (one dtype, no permutation, one token) AND the chain fills the device; per-layer otherwise."""
    lib = B.lib()

    def name(descs, tokens=1, flags=0):
        arr = (B.LayerDesc * len(descs))(*descs)
        return lib.vptq_quant_gemv_chain_kernel_name(arr, len(descs), tokens, flags)
    big, small = _canonical_desc(8192, 8192), _canonical_desc(4096, 4096)
    assert name([big] * 32) == b"gemv_k256c_kernel"
    assert name([big] * 32, flags=B.GEMV_CHAIN_DEPENDENT) == b"gemv_k256c_kernel"
    assert name([small] * 32) == b"gemv_k256c_kernel"          # 64 row groups each: 8 layers side by side
    assert name([big] * 9) == b"gemv_k256c_kernel"
    # round 4: up to 8 independent layers = ONE grouped launch of the one-layer kernels (faster than the persistent
    # launch there, profiles/r03/chain_vs_grouped_by_length.txt), also where the persistent kernel could not fill the device
    assert name([big] * 2) == name([big] * 8) == b"grouped"
    assert name([big] * 2, flags=B.GEMV_FORCE_MFMA) == b"gemv_k256c_kernel"   # (tests pin the persistent kernel)
    assert name([big] * 2, flags=B.GEMV_CHAIN_DEPENDENT) == b"gemv_k256c_kernel"
    assert name([_canonical_desc(8192, 1024)] * 3) == b"grouped"             # k / v sized projections
    assert name([big] * 4, tokens=2) == b"grouped"
    assert name([big] * 7 + [_canonical_desc(8192, 8192, perm=True)]) == b"grouped"
    assert name([small]) == b"per-layer"                       # one layer: the one-layer kernels
    assert name([small], flags=B.GEMV_FORCE_MFMA) == b"gemv_k256c_kernel"    # ... unless asked for
    assert name([big] * 32, tokens=2) == b"grouped"            # (any list of <= 64 layers the persistent kernel does not take)
    # (round 4: a layer with an input permutation stays in the persistent launch - x[perm] is gathered into the workspace
    # vptq_quant_gemv_chain_workspace_bytes_for asks for by a small launch in front of it)
    assert name([big] * 31 + [_canonical_desc(8192, 8192, perm=True)]) == b"gemv_k256c_kernel"
    assert lib.vptq_quant_gemv_chain_workspace_bytes_for((B.LayerDesc * 2)(big, _canonical_desc(8192, 8192, perm=True)), 2, 0) == 16384
    assert lib.vptq_quant_gemv_chain_workspace_bytes_for((B.LayerDesc * 2)(big, big), 2, 0) == 0
    assert name([big] * 31 + [_canonical_desc(8192, 8192, perm=True)], flags=B.GEMV_CHAIN_DEPENDENT) == b"per-layer"
    assert name([big] * 31 + [_canonical_desc(8192, 8192, dtype=1)]) == b"per-layer"   # mixed dtypes
    assert name([_canonical_desc(8192, 1024)] * 3, flags=B.GEMV_CHAIN_DEPENDENT) == b"per-layer"
    # the reference's roundings inside the chain launch (round 4): fp16, independent layers
    assert name([big] * 32, flags=B.GEMV_EXACT) == b"gemv_k256c_kernel"
    assert name([big] * 32, flags=B.GEMV_EXACT | B.GEMV_CHAIN_DEPENDENT) == b"per-layer"
    assert name([_canonical_desc(8192, 8192, dtype=1)] * 32, flags=B.GEMV_EXACT) == b"gemv_k256c_kernel"   # (bf16 since the end of round 6)
    assert name([big] * 40) == b"gemv_k256c_kernel"            # two launches (32 + 8)
    assert lib.vptq_quant_gemv_chain_workspace_bytes(32, 0) == 0
    assert lib.vptq_quant_gemv_chain_workspace_bytes(32, B.GEMV_CHAIN_DEPENDENT) == 32 * 1024


def test_kernel_choice_without_gpu():
    """Which kernel a call would use is host logic (vptq_quant_gemv_kernel_name launches nothing):
    the persistent MFMA kernel from 144 row groups on (bf16: 32), 1-4 tokens in its folded form,
    the exact form and small launches on the VALU kernel; the fused path up to 16 tokens."""
    if os.environ.get("VPTQ_K256_KERNEL"):
        pytest.skip("the kernel choice is pinned by VPTQ_K256_KERNEL")
    lib = B.lib()
    name = lambda d, tok, fl=0: lib.vptq_quant_gemv_kernel_name(d, tok, fl)  # noqa: E731
    big, small = _canonical_desc(8192, 8192), _canonical_desc(4096, 4096)
    assert lib.vptq_quant_gemv_max_tokens(big) == 48
    bbig = _canonical_desc(8192, 8192, dtype=1)
    assert lib.vptq_quant_gemv_max_tokens(bbig) == 48   # bf16 (round 3): the one-pass batched-decode kernel
    for tok in (1, 2, 3, 4):
        assert name(big, tok) == b"gemv_k256m_kernel<fast>"
        assert name(small, tok) == (b"gemv_k256_kernel<fast>" if tok <= 2 else b"gemv_k256_kernel")
    assert name(big, 1, B.GEMV_EXACT) == b"gemv_k256m_kernel"
    # several tokens in the reference's roundings (round 6): 2 token slots where they fit into LDS, 4 slots up to 2 (fp16) /
    # 1 (bf16) sweeps of 2048 columns, from 144 row groups on - else the VALU kernel
    assert name(big, 2, B.GEMV_EXACT) == b"gemv_k256m_kernel"
    assert name(big, 3, B.GEMV_EXACT) == name(big, 4, B.GEMV_EXACT) == b"gemv_k256_kernel"     # 4 sweeps: 4 slots would spill
    assert name(small, 2, B.GEMV_EXACT) == b"gemv_k256_kernel"                                # 128 row groups
    tall, btall = _canonical_desc(4096, 14336), _canonical_desc(2048, 8192, dtype=1)
    assert name(tall, 2, B.GEMV_EXACT) == name(tall, 4, B.GEMV_EXACT) == b"gemv_k256m_kernel"  # 2 sweeps, fp16
    assert name(btall, 2, B.GEMV_EXACT) == name(btall, 3, B.GEMV_EXACT) == b"gemv_k256m_kernel"  # 1 sweep, bf16
    assert name(bbig, 2, B.GEMV_EXACT) == b"gemv_k256m_kernel" and name(bbig, 4, B.GEMV_EXACT) == b"gemv_k256_kernel"
    assert name(_canonical_desc(4096, 4096, dtype=1), 2, B.GEMV_EXACT) == b"gemv_k256_kernel"  # bf16 too: 144 row groups for several tokens
    assert name(_canonical_desc(4096, 4096, dtype=1), 1, B.GEMV_EXACT) == b"gemv_k256m_kernel"  # (one token: from 32 on, matrix-pipe roundings)
    assert name(_canonical_desc(14336, 8192), 1, B.GEMV_EXACT) == b"gemv_k256m_kernel"         # 7 sweeps: scale and bias staged in LDS
    assert name(big, 1, B.GEMV_FORCE_VALU) == b"gemv_k256_kernel<fast>"
    assert name(small, 1, B.GEMV_FORCE_MFMA) == b"gemv_k256m_kernel<fast>"
    assert name(big, 1, B.GEMV_FORCE_GENERIC) == b"gemv_generic_kernel"
    # 5+ tokens: ONE pass over the indices (gemm_k256t, launches of 16; wants the workspace vptq_quant_gemv_workspace_bytes
    # names); with the reference's roundings: gemm_k256 (fp16)
    assert name(big, 4) == b"gemv_k256m_kernel<fast>" and name(big, 5) == name(big, 16) == b"gemm_k256t_kernel"
    assert name(big, 64) == b"gemm_k256t_kernel" and name(big, 65) is None          # launches of 16 tokens
    assert name(big, 5, B.GEMV_EXACT) == name(big, 64, B.GEMV_EXACT) == b"gemm_k256_kernel"
    # bf16 likewise (round 6: the batched kernel with the reference's roundings serves bf16 too)
    assert name(bbig, 4) == b"gemv_k256m_kernel<fast>" and name(bbig, 5) == name(bbig, 16) == name(bbig, 64) == b"gemm_k256t_kernel"
    assert name(bbig, 65) is None and name(bbig, 5, B.GEMV_EXACT) == name(bbig, 64, B.GEMV_EXACT) == b"gemm_k256_kernel"
    assert lib.vptq_quant_gemv_workspace_bytes(bbig, 5, 0) == 64 * 4096 + 256 and lib.vptq_quant_gemv_workspace_bytes(bbig, 4, 0) == 0
    assert lib.vptq_quant_gemv_workspace_bytes(big, 16, 0) == 64 * 4096 + 256 and lib.vptq_quant_gemv_workspace_bytes(big, 16, B.GEMV_EXACT) == 0
    assert name(big, 2, B.GEMV_FORCE_BATCHED) == name(big, 16, B.GEMV_FORCE_BATCHED) == b"gemm_k256t_kernel"
    assert lib.vptq_quant_gemv_workspace_bytes(big, 2, B.GEMV_FORCE_BATCHED) == 64 * 4096 + 256
    # bf16: folded form in the MFMA kernel from 32 row groups (128 vector-rows) on
    assert name(_canonical_desc(4096, 1024, dtype=1), 1) == b"gemv_k256m_kernel<fast>"
    assert name(_canonical_desc(4096, 512, dtype=1), 1) == b"gemv_k256_kernel"
    # token slots that do not fit beside the codebook image / columns that cannot be staged
    wide = _canonical_desc(14336, 8192)
    assert name(wide, 2) == b"gemv_k256m_kernel<fast>" and name(wide, 4) == b"gemv_k256_kernel"
    wider = _canonical_desc(28672, 8192)
    assert name(wider, 1) == b"gemv_k256m_kernel<fast>" and name(wider, 2) == b"gemv_k256_kernel<fast>"
    assert name(_canonical_desc(28672, 8192, perm=True), 1) == b"gemv_k256_kernel<fast>"
    # grouped: the decision is taken for the whole launch
    descs = (B.LayerDesc * 3)(_canonical_desc(4096, 4096), _canonical_desc(4096, 1024),
                              _canonical_desc(4096, 1024))
    assert lib.vptq_quant_gemv_grouped_kernel_name(descs, 3, 1, 0) == b"gemv_k256m_kernel<fast>"  # 192 row groups
    assert lib.vptq_quant_gemv_grouped_kernel_name(descs, 2, 1, 0) == b"gemv_k256m_kernel<fast>"  # 160
    two = (B.LayerDesc * 2)(_canonical_desc(4096, 1024), _canonical_desc(4096, 1024))
    assert lib.vptq_quant_gemv_grouped_kernel_name(two, 2, 1, 0) == b"gemv_k256_kernel<fast>"     # 64


def _format_desc(v, k, kr, I=4096, O=4096, C=1, outliers=0, norm=True):
    """A descriptor of an arbitrary index format with fake aligned pointers."""
    import math
    d = B.LayerDesc()
    ib, rb = int(math.log2(k)), (int(math.log2(kr)) if kr else 0)
    G = (I - outliers) // C
    d.in_features, d.out_features, d.vector_len, d.num_codebooks, d.group_size = I, O, v, C, G
    d.num_centroids, d.num_res_centroids, d.index_bits, d.res_bits = k, kr, ib, rb
    d.row_words, d.num_indices, d.dtype = (G * (ib + rb) + 31) // 32, (O + v - 1) // v, 0
    d.indices, d.centroids = 1 << 20, 2 << 20
    d.res_centroids = (3 << 20) if kr else None
    if norm:
        d.weight_scale, d.weight_bias = 4 << 20, 5 << 20
    if outliers:
        d.outlier_size, d.outlier_vector_len, d.num_outlier_centroids = outliers, v, 1024
        d.num_outlier_indices = d.num_indices
        d.outlier_indices, d.outlier_centroids = 6 << 20, 7 << 20
    return d


def test_format_routing_without_gpu():
    """Which kernel serves which index format (host logic only): canonical -> k = 256 kernels, v = 8 with
    k = 65536 and 16 / 24 / 32 index bits -> gather, k <= 8192 + kr <= 512 -> LDS-resident, the other
    formats of every vector length (outlier columns of the same vector length, or of vector length 4 under
    v = 8 / 12 / 16, included) -> gatherx, the rest -> generic."""
    lib = B.lib()
    name = lambda d, tok=1, fl=0: lib.vptq_quant_gemv_kernel_name(d, tok, fl)  # noqa: E731
    assert name(_format_desc(8, 256, 256)).startswith(b"gemv_k256")
    assert name(_format_desc(8, 65536, 0)) == name(_format_desc(8, 65536, 256)) == \
        name(_format_desc(8, 65536, 65536)) == b"gemv_gather_kernel"
    assert name(_format_desc(8, 8192, 256)) == name(_format_desc(8, 4096, 512)) == b"gemv_lds_kernel"
    # ... one token in the default arithmetic on >= 1024 vector-rows: its MFMA variant (folded form)
    big = _format_desc(8, 8192, 256, I=8192, O=8192)
    assert name(big) == b"gemv_lds_mfma_kernel" and name(big, 2) == name(big, 1, B.GEMV_EXACT) == b"gemv_lds_kernel"
    assert name(_format_desc(8, 8192, 256, I=16384, O=8192)) == b"gemv_lds_kernel"   # no LDS left for the activations
    for v, k, kr in ((16, 65536, 65536), (16, 65536, 32768), (16, 65536, 1024), (16, 65536, 0),
                     (12, 65536, 4096), (8, 65536, 1024), (8, 32768, 0), (8, 16384, 16384), (8, 8192, 1024),
                     (16, 256, 256), (8, 256, 0)):
        d = _format_desc(v, k, kr, I=4096, O=v * 512)
        for tok in (1, 2, 3, 4, 5, 8):
            assert name(d, tok) == b"gemv_gatherx_kernel", (v, k, kr, tok)
        assert name(d, 1, B.GEMV_FORCE_GENERIC) == b"gemv_generic_kernel"
        assert name(d, 16) == b"gemv_gatherx_kernel" and name(d, 17) is None   # launches of <= 4 tokens
        assert lib.vptq_quant_gemv_max_tokens(d) == 8                           # beyond: dequant + GEMM is faster
    assert name(_format_desc(8, 4096, 4096, C=2)) == b"gemv_gatherx_kernel"          # two codebook groups
    assert name(_format_desc(8, 1024, 4, norm=False)) == b"gemv_lds_kernel"
    assert name(_format_desc(16, 1024, 4, O=16 * 64, norm=False)) == b"gemv_gatherx_kernel"   # no norm
    # outlier columns
    assert name(_format_desc(8, 65536, 256, outliers=128)) == b"gemv_gatherx_kernel"   # same vector length
    assert name(_format_desc(16, 65536, 65536, O=16 * 256, outliers=64)) == b"gemv_gatherx_kernel"
    mixed = _format_desc(8, 65536, 256, outliers=128)
    mixed.outlier_vector_len, mixed.num_outlier_indices = 4, 4096 // 4
    assert name(mixed) == b"gemv_gatherx_kernel"          # vector length 4 (the reference kernel's only one) under 8
    for v in (12, 16):
        m4 = _format_desc(v, 65536, 4096, O=v * 128, outliers=64)
        m4.outlier_vector_len, m4.num_outlier_indices = 4, v * 128 // 4
        assert name(m4) == b"gemv_gatherx_kernel"
    m6 = _format_desc(6, 4096, 0, O=6 * 128, outliers=64)
    m6.outlier_vector_len, m6.num_outlier_indices = 4, 6 * 128 // 4
    assert name(m6) == b"gemv_generic_kernel"             # outlier rows straddle the vector-rows of length 6
    m2 = _format_desc(8, 65536, 256, outliers=128)
    m2.outlier_vector_len, m2.num_outlier_indices = 2, 4096 // 2
    assert name(m2) == b"gemv_generic_kernel"             # another outlier vector length
    # column counts that are no multiple of 4: generic
    assert name(_format_desc(8, 65536, 256, I=4096 + 2, outliers=130)) == b"gemv_generic_kernel"
    for v in (2, 4, 6, 10):                               # every vector length the reference dispatches
        for k, kr in ((4096, 0), (65536, 256), (256, 256)):
            assert name(_format_desc(v, k, kr, O=v * 256)) == b"gemv_gatherx_kernel", (v, k, kr)
    assert name(_format_desc(16, 65536, 0, I=4098, O=16 * 64)) == b"gemv_generic_kernel"


def test_validation_errors_without_gpu():
    """The C ABI validates before launching: exercisable with no device."""
    lib = B.lib()
    d = B.LayerDesc()
    rc = lib.vptq_quant_gemv(d, None, None, 1, 0, None, 0, None)
    assert rc < 0 and b"NULL" in lib.vptq_last_error()
    d.indices, d.centroids = 16, 16   # fake non-null pointers; nothing is dereferenced
    d.dtype = 7
    assert lib.vptq_quant_gemv(d, 16, 16, 1, 0, None, 0, None) == -3
    d.dtype = 0
    d.vector_len = 7
    assert lib.vptq_quant_gemv(d, 16, 16, 1, 0, None, 0, None) == -3
    d.vector_len, d.in_features, d.out_features = 8, 64, 64
    d.num_codebooks, d.group_size = 1, 32           # 32 != 64
    assert lib.vptq_quant_gemv(d, 16, 16, 1, 0, None, 0, None) == -2
    d.group_size = 64
    d.num_centroids, d.index_bits = 256, 8
    d.row_words, d.num_indices = 16, 8
    assert lib.vptq_quant_gemv(d, 16, 16, 0, 0, None, 0, None) == -5      # tokens
    assert lib.vptq_quant_gemv(d, 16, 16, 99, 0, None, 0, None) == -5
    d.row_words = 3
    assert lib.vptq_quant_gemv(d, 16, 16, 1, 0, None, 0, None) == -2      # row too short
    # 4 GiB or more of packed indices: the kernels' 32-bit byte offsets would wrap - refused, loudly
    d.out_features, d.num_indices, d.row_words = 8 * (1 << 20), 1 << 20, 1024
    assert lib.vptq_quant_gemv(d, 16, 16, 1, 0, None, 0, None) == -2 and b"4 GiB" in lib.vptq_last_error()
    d.row_words = 1023                                                     # just under: accepted by the validation
    assert lib.vptq_quant_gemv_kernel_name(d, 1, 0) is not None
    d.out_features, d.num_indices = 64, 8
    d.row_words = 16
    d.perm = 16
    assert lib.vptq_dequant(d, 16, None) == -1                             # needs inv_perm
    assert lib.vptq_quant_gemv_kernel_name(d, 1, 0) == b"gemv_gatherx_kernel"   # v = 8, no outliers
    assert lib.vptq_quant_gemv_kernel_name(d, 1, B.GEMV_FORCE_GENERIC) == b"gemv_generic_kernel"
    d.vector_len, d.num_indices = 4, 16
    assert lib.vptq_quant_gemv_kernel_name(d, 1, 0) == b"gemv_gatherx_kernel"   # v = 4
    d.group_size = d.in_features = 66
    d.row_words = 17
    assert lib.vptq_quant_gemv_kernel_name(d, 1, 0) == b"gemv_generic_kernel"   # 66 columns: no whole pieces of 4
    d.group_size = d.in_features = 64
    d.row_words = 16
    d.vector_len, d.num_indices = 8, 8
    assert lib.vptq_quant_gemv_max_tokens(d) == 8                          # not the canonical format
    assert lib.vptq_quant_gemv_max_tokens(B.LayerDesc()) == 0
    v2 = B.V2Desc()
    assert lib.vptq_quant_gemv_v2(v2, 16, 16, 1, 0, None) == -1
    v2.indices = v2.centroids = 16
    v2.vector_len, v2.in_features, v2.out_features, v2.num_centroids = 8, 32, 64, 8192
    assert lib.vptq_quant_gemv_v2(v2, 16, 16, 16, 0, None) == -5           # tokens < 16


def test_cpu_tensors_raise_no_silent_fallback():
    L, x, y, cfg, _ = load_golden("canon_k256x2")
    m = spec_to_module(L, "cpu")
    xt = torch.zeros(1, 1, L.in_features, dtype=torch.float16)
    with pytest.raises(RuntimeError, match="no CPU fallback|ROCm devices only"):
        m(xt)
    with pytest.raises(RuntimeError):
        m.dequant()


@pytest.mark.parametrize("ib,rb,G", [(8, 8, 64), (12, 0, 7), (13, 8, 33), (16, 8, 5),
                                     (16, 16, 3), (4, 2, 1), (10, 4, 96), (12, 12, 128)])
def test_pack_tools_match_oracle(ib, rb, G):
    rng = np.random.default_rng(ib * 100 + rb)
    idx = rng.integers(0, 1 << ib, size=(2, 3, G))
    r = rng.integers(0, 1 << rb, size=(2, 3, G)) if rb else None
    want = vo.pack_indices(idx, ib, r, rb)
    ti = torch.from_numpy(idx.astype(np.uint16).view(np.int16))
    tr = torch.from_numpy(r.astype(np.uint16).view(np.int16)) if rb else None
    got = pack_index(ti, ib, tr, rb)
    assert got.dtype == torch.int32 and (got.numpy() == want).all()
    a, b = unpack_index_tensor(got, ib, G, rb, G)
    assert (a.numpy() == idx).all()
    if rb:
        assert (b.numpy() == r).all()


@pytest.mark.parametrize("name", golden_names())
def test_module_state_dict_contract(name):
    """Names / shapes / dtypes are the on-disk contract of HF checkpoints."""
    L, x, y, cfg, _ = load_golden(name)
    m = spec_to_module(L, "cpu")
    sd = m.state_dict()
    I, O, v, C = L.in_features, L.out_features, L.vector_len, L.num_codebooks
    T = L.index_bits + L.res_bits
    assert sd["indices"].dtype == torch.int32
    assert tuple(sd["indices"].shape) == (C, L.num_indices, (L.group_size * T + 31) // 32)
    assert tuple(sd["centroids.weight"].shape) == (C, L.num_centroids * v)
    assert ("res_centroids.weight" in sd) == (L.num_res_centroids > 0)
    assert "res_indices" not in sd
    if L.perm is not None:
        assert sd["perm"].dtype == torch.int16 and tuple(sd["perm"].shape) == (I,)
    if L.enable_outlier:
        assert sd["outlier_indices"].dtype == torch.int16
        assert tuple(sd["outlier_centroids.weight"].shape) == (
            1, L.num_outlier_centroids * L.outlier_vector_len)
    if L.weight_scale is not None:
        assert tuple(sd["weight_scale"].shape) == (I,) and tuple(sd["weight_bias"].shape) == (I,)
    assert ("bias" in sd) == (L.bias is not None)


@pytest.mark.skipif(not reference_available(), reason="/root/reference not mounted")
def test_module_matches_reference_state_dict_and_meta_init():
    import gen_golden as gg
    from _refshim import load_reference
    ref = load_reference()
    for name, I, O, kw, dtype, tokens, dist in gg.CASES:
        C, S = kw["group_num"], kw["outlier_size"]
        G = (I - S) // C
        common = dict(group_size=G, indices_as_float=False, is_indice_packed=True,
                      dtype=gg.TORCH_DT[dtype], enable_proxy_error=False, **kw)
        rm = ref.VQuantLinear(I, O, **common)
        with torch.device("meta"):
            mm = vptq_amd.VQuantLinear(I, O, **common)
        rs, ms = rm.state_dict(), mm.state_dict()
        assert list(rs) == list(ms), name
        for k in rs:
            assert rs[k].shape == ms[k].shape and rs[k].dtype == ms[k].dtype, (name, k)
        for attr in ("padding", "num_indices", "outlier_padding", "vector_len", "num_codebooks",
                     "num_centroids", "num_res_centroids", "enable_outlier", "enable_residual"):
            assert getattr(rm, attr) == getattr(mm, attr), (name, attr)


def test_derived_state_cache_follows_the_tensor_not_the_address():
    """inv_perm / permuted scale are cached on the owning tensor object: a new tensor that
    happens to reuse a freed tensor's address must not see stale derived state."""
    for _ in range(20):
        perm = torch.randperm(64).to(torch.int32).to(torch.int16)
        t = torch.randn(64).half()
        assert torch.equal(B.inverse_perm(perm).long(), torch.argsort(perm.long()))
        assert torch.equal(B.permuted_norm(perm, t, "scale"), t[perm.long()])
        del perm, t
    perm = torch.randperm(64).to(torch.int32).to(torch.int16)
    first = B.inverse_perm(perm)
    assert B.inverse_perm(perm) is first                      # cached
    perm.data = torch.randperm(64).to(torch.int32).to(torch.int16)
    assert torch.equal(B.inverse_perm(perm).long(), torch.argsort(perm.long()))


def test_absorb_perm_matches_oracle_weight():
    """Folding perm into the index order leaves the dense weight unchanged (oracle)."""
    from vptq_amd.utils.pack import absorb_perm_layer
    L = vo.make_layer(256, 64, dist="llm", seed=5, enable_perm=True, num_centroids=4096,
                      num_res_centroids=256)
    W_before = vo.dequant(L)
    m = spec_to_module(L, "cpu")
    assert absorb_perm_layer(m) and m.enable_perm is False and m.perm is None
    assert "perm" not in m.state_dict()
    L2 = vo.LayerSpec(256, 64, 8, 4096, 256, 1, 256, dtype="f16")
    L2.indices = m.indices.numpy()
    L2.centroids, L2.res_centroids = L.centroids, L.res_centroids
    L2.weight_scale, L2.weight_bias = L.weight_scale, L.weight_bias
    assert (vo.dequant(L2) == W_before).all()
    assert absorb_perm_layer(m) is False          # idempotent: nothing left to absorb


def test_absorb_perm_keeps_wide_residual_indices():
    """More residual than main centroids (res_bits > index_bits): the reference's Python unpack
    masks the residual with index_bits (pack.py:137); the tool that REWRITES indices must not."""
    from vptq_amd.utils.pack import absorb_perm_layer
    L = vo.make_layer(128, 32, dist="llm", seed=11, enable_perm=True, num_centroids=16,
                      num_res_centroids=256)
    W_before = vo.dequant(L, ref_residual_mask_quirk=False)   # the kernels' (res_bits) mask
    m = spec_to_module(L, "cpu")
    assert absorb_perm_layer(m)
    L2 = vo.LayerSpec(128, 32, 8, 16, 256, 1, 128, dtype="f16")
    L2.indices = m.indices.numpy()
    L2.centroids, L2.res_centroids = L.centroids, L.res_centroids
    L2.weight_scale, L2.weight_bias = L.weight_scale, L.weight_bias
    assert (vo.dequant(L2, ref_residual_mask_quirk=False) == W_before).all()


def test_tensor_version_tolerates_inference_tensors():
    t = torch.zeros(4)
    v0 = B.tensor_version(t)
    t.add_(1)
    assert B.tensor_version(t) == v0 + 1 and B.tensor_version(None) == 0
    with torch.inference_mode():
        u = torch.zeros(4)
    assert B.tensor_version(u) == -1
    perm = torch.randperm(32).to(torch.int32).to(torch.int16)
    with torch.inference_mode():
        pi = perm.clone()
        assert torch.equal(B.inverse_perm(pi).long(), torch.argsort(pi.long()))


def test_vptq_alias_for_hf():
    import sys
    # other tests may have imported the REFERENCE under the name `vptq`
    saved = {k: sys.modules.pop(k) for k in list(sys.modules)
             if k == "vptq" or k.startswith("vptq.")}
    try:
        _check_alias()
    finally:
        for k in [k for k in sys.modules if k == "vptq" or k.startswith("vptq.")]:
            del sys.modules[k]
        sys.modules.update(saved)


def _check_alias():
    import vptq
    assert vptq.__file__.startswith(ROOT)
    assert vptq.VQuantLinear is vptq_amd.VQuantLinear
    from packaging.version import Version
    assert Version(vptq.__version__) >= Version("0.0.4")
    import vptq.ops
    assert vptq.ops.quant_gemm is vptq_amd.ops.quant_gemm
    assert hasattr(vptq.ops, "dequant") and hasattr(vptq.ops, "quant_gemv_v2")
    # the reference's canonical LEAF import paths give the same objects (not a second copy of
    # the module whose classes fail isinstance checks)
    from vptq.layers.vqlinear import VQuantLinear as leaf_cls
    from vptq.layers.model_base import AutoModelForCausalLM as leaf_auto
    from vptq.ops.quant_gemm import quant_gemm as leaf_op
    from vptq.utils.pack import pack_index as leaf_pack
    import importlib
    importlib.import_module("vptq_amd.utils.pack")
    assert leaf_cls is vptq_amd.VQuantLinear and leaf_auto is vptq_amd.AutoModelForCausalLM
    assert leaf_op is vptq_amd.ops.quant_gemm and leaf_pack is vptq_amd.utils.pack.pack_index
    import sys
    assert sys.modules["vptq.layers.vqlinear"] is sys.modules["vptq_amd.layers.vqlinear"]


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: the package must not reference it."""
    pkg = os.path.join(ROOT, "vptq_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dp, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f


def test_command_line_options_match_the_reference():
    """`python -m vptq`: same options and defaults as the reference's parser (vptq/app_utils.py:17-53);
    compared live when the reference is mounted, against its documented set otherwise."""
    import importlib.util
    import sys
    import vptq_amd.app_utils as app
    # other tests may have imported the REFERENCE under the name `vptq`
    saved = {k: sys.modules.pop(k) for k in list(sys.modules) if k == "vptq" or k.startswith("vptq.")}
    try:
        import vptq
        assert vptq.__file__.startswith(ROOT)
        assert vptq.app_utils is app and importlib.util.find_spec("vptq.__main__") is not None
    finally:
        for k in [k for k in sys.modules if k == "vptq" or k.startswith("vptq.")]:
            del sys.modules[k]
        sys.modules.update(saved)
    want = {"model": None, "tokenizer": "", "prompt": "once upon a time, there ", "chat": False,
            "chat_system_prompt": "you are a math teacher."}
    ours = {a.dest: a.default for a in app.define_basic_args()._actions if a.dest != "help"}
    assert ours == want
    ref_file = "/root/reference/vptq/app_utils.py"
    if os.path.exists(ref_file):
        import re
        src = open(ref_file).read()
        flags = set(re.findall(r'"(--[a-z-]+)"', src))
        mine = {s for a in app.define_basic_args()._actions for s in a.option_strings if s.startswith("--")}
        assert flags == mine - {"--help"}, (flags, mine)
    ns = app.define_basic_args().parse_args(["--model", "m", "--chat", "--chat-system-prompt", "s"])
    assert ns.model == "m" and ns.chat and ns.chat_system_prompt == "s" and ns.tokenizer == ""
    with pytest.raises(SystemExit):
        app.define_basic_args().parse_args([])      # --model is required
    for fn in ("define_basic_args", "eval_prompt", "chat_loop", "get_chat_loop_generator", "get_valid_args", "main"):
        assert callable(getattr(app, fn))


def test_command_line_chat_loop_logic_without_gpu(capsys):
    """chat_loop / eval_prompt with stand-ins for model and tokenizer: turns alternate, the reply is what follows
    the prompt tokens, `exit` / an empty line end the chat, no chat template -> prompt completion (reference
    vptq/app_utils.py:65-110)."""
    import types
    import vptq_amd.app_utils as app

    class Tok:
        chat_template = "x"

        def __call__(self, text, return_tensors=None):
            class Enc(dict):
                def to(self, dev):
                    return self
            return Enc(input_ids=torch.tensor([[1, 2, 3]]))

        def apply_chat_template(self, messages, add_generation_prompt=True, return_tensors=None):
            self.last = [dict(m) for m in messages]
            return torch.arange(len(messages) * 2).reshape(1, -1)

        def batch_decode(self, ids, skip_special_tokens=True):
            return ["reply-%d" % ids.shape[-1]]

        def decode(self, *a, **k):
            return ""

    class Model:
        device = torch.device("cpu")

        def generate(self, ids=None, streamer=None, max_new_tokens=0, **kw):
            self.kw = dict(kw, max_new_tokens=max_new_tokens)
            base = ids if ids is not None else kw["input_ids"]
            return torch.cat([base, torch.zeros(1, 3, dtype=base.dtype)], dim=1)

    tok, model = Tok(), Model()
    args = app.define_basic_args().parse_args(["--model", "m", "--chat", "--chat-system-prompt", "sys"])
    lines = iter(["hello", "again", ""])
    hist = app.chat_loop(model, tok, args, read=lambda p: next(lines))
    assert [m["role"] for m in hist] == ["system", "user", "assistant", "user", "assistant"]
    assert hist[0]["content"] == "sys" and hist[1]["content"] == "hello" and hist[2]["content"] == "reply-3"
    assert model.kw["max_new_tokens"] == 500 and model.kw["do_sample"] is True and model.kw["pad_token_id"] == 2
    assert tok.last[-1] == {"role": "user", "content": "again"}
    # prompt completion: 100 greedy tokens
    args2 = app.define_basic_args().parse_args(["--model", "m", "--prompt", "p"])
    out = app.chat_loop(model, tok, args2)
    assert out.shape == (1, 6) and model.kw["max_new_tokens"] == 100 and "do_sample" not in model.kw
    # no chat template: falls back to the prompt completion with a warning
    tok.chat_template = None
    out = app.chat_loop(model, tok, args)
    assert out.shape == (1, 6) and "no chat_template" in capsys.readouterr().out


def test_chat_generator_streams_from_a_worker_thread(capsys):
    """the UI callback (reference vptq/app_utils.py:109-165, imported by vptq/app.py:17): generate() runs on a worker
    thread and the reply arrives piece by piece through a TextIteratorStreamer; sampling options are passed on; a
    tokenizer without a chat template is refused with the reference's error"""
    import threading
    import vptq_amd.app_utils as app

    class Enc(dict):
        def to(self, dev):
            return self

    class Tok:
        chat_template = "x"

        def apply_chat_template(self, messages, add_generation_prompt=True, return_tensors=None, return_dict=False):
            assert return_dict and add_generation_prompt
            return Enc(input_ids=torch.tensor([[5, 6]]), attention_mask=torch.ones(1, 2, dtype=torch.long))

        def decode(self, ids, **kw):
            return "".join("w%d " % int(i) for i in ids)

    class Model:
        device = torch.device("cpu")

        def generate(self, input_ids=None, streamer=None, **kw):
            self.kw, self.thread = kw, threading.current_thread()
            streamer.put(input_ids)                      # the prompt (skipped by the streamer)
            for t in (7, 8, 9):
                streamer.put(torch.tensor([t]))
            streamer.end()

    model = Model()
    gen = app.chat_generator(model, Tok())
    pieces = list(gen([{"role": "user", "content": "hi"}], 12, temperature=0.5, top_p=0.9))
    assert "".join(pieces).split() == ["w7", "w8", "w9"]
    assert model.thread is not threading.current_thread()
    assert model.kw["max_new_tokens"] == 12 and model.kw["do_sample"] is True and model.kw["pad_token_id"] == 2
    assert model.kw["temperature"] == 0.5 and model.kw["top_p"] == 0.9 and "attention_mask" in model.kw
    assert "Press 'exit' to quit" in capsys.readouterr().out
    tok = Tok()
    tok.chat_template = None
    with pytest.raises(Exception, match="chat_template"):
        app.chat_generator(model, tok)


def test_module_copies_and_pickles_without_its_derived_state():
    """descriptors (ctypes), the sliced layout and sibling links live in the module's __dict__ and are device-bound:
    copy.deepcopy / pickle carry the parameters and the configuration only; the derived state is rebuilt on demand"""
    import copy, pickle
    import vptq_amd
    m = vptq_amd.VQuantLinear(64, 32, vector_lens=[-1, 8], num_centroids=[-1, 256], num_res_centroids=[-1, 256], group_num=1,
                              group_size=64, outlier_size=0, indices_as_float=False, enable_norm=True, enable_perm=False,
                              is_indice_packed=True, bias=True, dtype=torch.float16, device="cpu", enable_proxy_error=False)
    m.__dict__["_desc_cache"] = ("stand-in", object())
    m.__dict__["_sliced"] = (1, object())
    m.__dict__["_sliced_cand"] = True
    m.enable_sliced_layout(False)
    m2 = copy.deepcopy(m)
    assert "_desc_cache" not in m2.__dict__ and "_sliced" not in m2.__dict__ and m2.__dict__["_sliced_on"] is False
    assert m2.indices is not m.indices and torch.equal(m2.indices, m.indices)
    m.__dict__["_sliced"] = (1, object())
    m3 = pickle.loads(pickle.dumps(m))
    assert "_desc_cache" not in m3.__dict__ and "_sliced" not in m3.__dict__
    assert sorted(m3.state_dict()) == sorted(m.state_dict())


def _family_desc(I, O, v, k, kr):
    """descriptor of a large-codebook layer (fake aligned pointers, never dereferenced)"""
    d = B.LayerDesc()
    ib, rb = k.bit_length() - 1, (kr.bit_length() - 1 if kr else 0)
    d.in_features, d.out_features, d.vector_len, d.num_codebooks, d.group_size = I, O, v, 1, I
    d.num_centroids, d.num_res_centroids, d.index_bits, d.res_bits = k, kr, ib, rb
    d.row_words, d.num_indices, d.dtype = (I * (ib + rb) + 31) // 32, (O + v - 1) // v, 0
    d.indices, d.centroids, d.res_centroids = 1 << 20, 2 << 20, (3 << 20 if kr else None)
    d.weight_scale, d.weight_bias = 4 << 20, 5 << 20
    return d


def test_sliced_layout_routing_without_gpu():
    """which layers get sliced layouts, how many tables, which slice counts, whole-table mode: host logic"""
    lib = B.lib()
    sup, tabs, whole = lib.vptq_sliced_layout_supported, lib.vptq_sliced_layout_tables, lib.vptq_sliced_layout_whole_table
    assert sup(_family_desc(8192, 8192, 8, 65536, 0)) == 8 and tabs(_family_desc(8192, 8192, 8, 65536, 0)) == 1
    assert sup(_family_desc(8192, 8192, 8, 65536, 256)) == 8 and tabs(_family_desc(8192, 8192, 8, 65536, 256)) == 1   # byte side stream
    assert sup(_family_desc(14336, 4096, 8, 65536, 256)) == 16                     # the 4 KiB table no longer fits beside 8 slices
    assert sup(_family_desc(28672, 8192, 8, 65536, 65536)) == 16 and tabs(_family_desc(28672, 8192, 8, 65536, 65536)) == 2
    assert sup(_family_desc(8192, 8192, 16, 65536, 65536)) == 16 and sup(_family_desc(28672, 8192, 16, 65536, 0)) == 32
    assert tabs(_family_desc(8192, 8192, 16, 65536, 256)) == 2                     # v = 16 has no byte side stream
    assert sup(_family_desc(8192, 8192, 8, 32768, 0)) == 8 and sup(_family_desc(8192, 8192, 8, 8192, 0)) == 0   # (k <= 8192: gemv_lds)
    assert sup(_family_desc(8192, 8192, 12, 65536, 0)) == 0 and sup(_family_desc(40000, 8192, 8, 65536, 0)) == 0
    # a small second table is held whole ... while it fits beside the activations
    assert whole(_family_desc(8192, 8192, 16, 65536, 1024), 1) == 1 and whole(_family_desc(8192, 8192, 16, 65536, 1024), 0) == 0
    assert whole(_family_desc(8192, 8192, 16, 65536, 65536), 1) == 0               # 128 KiB slices
    assert whole(_family_desc(8192, 8192, 8, 65536, 4), 1) == 1
    assert whole(_family_desc(25704, 768, 16, 65536, 4096), 1) == 0                # 128 KiB of table + 51 KiB of activations: sliced
    assert sup(_family_desc(25704, 768, 16, 65536, 4096)) == 32


def test_exact_sliced_layout_routing_without_gpu():
    """ABI 8: the slice count a layout needs for the reference's roundings (scale, bias and x of every column beside the slice: 6
    bytes of LDS per column), which layers that arithmetic serves over a layout, and what the entry point turns down: host logic"""
    import ctypes as C
    lib = B.lib()
    EX = B.GEMV_EXACT
    supf = lib.vptq_sliced_layout_supported_for

    def want(I, v, k, kr):       # the LDS budget restated: 160 KiB = table slice + 6 (I + 64) + 64 (+ 4 KiB for v8's 256-entry table)
        small = 8 if v == 8 else 16
        for nsl in (small, 2 * small):
            if (k // nsl) * v * 2 + (I + 64) * 6 + 64 + (4096 if (v == 8 and kr == 256) else 0) <= 163840:
                return nsl
        return 0
    for (I, v, k, kr) in ((4096, 8, 65536, 0), (4096, 8, 65536, 256), (4704, 8, 65536, 256), (4712, 8, 65536, 256), (5376, 8, 65536, 0),
                          (5384, 8, 65536, 0), (8192, 8, 65536, 256), (14336, 8, 65536, 256), (16288, 8, 65536, 0), (16296, 8, 65536, 0),
                          (28672, 8, 65536, 0), (8192, 16, 65536, 0), (4096, 16, 65536, 0), (14336, 8, 16384, 64), (8192, 8, 65536, 65536),
                          (8192, 16, 65536, 65536), (8192, 8, 65536, 4096)):
        d = _family_desc(I, 4096, v, k, kr)
        assert supf(d, EX) == want(I, v, k, kr), (I, v, k, kr, supf(d, EX))
        assert supf(d, 0) == lib.vptq_sliced_layout_supported(d)
        assert supf(d, EX | B.GEMV_FORCE_GENERIC) == 0
    assert supf(_family_desc(8192, 8192, 8, 8192, 0), EX) == 0 and supf(_family_desc(8192, 8192, 12, 65536, 0), EX) == 0
    # the entry point: a layout with the folded form's slice count, or without the residual side stream, is "unsupported"
    buf = (C.c_char * 64)()
    p = C.addressof(buf)
    d = _family_desc(8192, 8192, 8, 65536, 256)
    lay8 = (B.SlicedLayout * 1)(B.SlicedLayout(p, p, p, p, 2, 1, 8, 0, None))
    lay16 = (B.SlicedLayout * 1)(B.SlicedLayout(p, p, p, None, 2, 1, 16, 0, None))
    x = (C.c_char * 64)()
    xa = (C.addressof(x) + 15) & ~15
    ws = 1 << 24   # (never dereferenced: the call is turned down before any launch)
    assert lib.vptq_quant_gemv_sliced(d, lay8, xa, xa, EX, xa, ws, None) == B.E_UNSUPPORTED      # 8 slices: the folded layout
    assert lib.vptq_quant_gemv_sliced(d, lay16, xa, xa, EX, xa, ws, None) == B.E_UNSUPPORTED     # no residual side stream
    assert b"res" in lib.vptq_last_error()


def test_exact_column_parts_without_gpu():
    """layers too wide for the reference's roundings in one piece (6 bytes of LDS per column beside the slice) are served as 2 or 3
    equal column parts of a multiple of 8 columns (VPTQ_GEMV_COLUMN_PARTS): which layers, which descriptors - host logic"""
    from vptq_amd.utils.sliced import exact_column_parts, part_desc
    for (I, v, kr, want) in ((8192, 8, 256, (1, 16)), (4096, 8, 0, (1, 8)), (28672, 8, 256, (2, 16)), (28672, 8, 0, (2, 16)), (24576, 8, 65536, (2, 16)),
                             (16392, 8, 0, (3, 16)), (32768, 8, 0, (0, 0)), (28672, 16, 0, (2, 32)), (16304, 8, 0, (1, 16)), (16312, 8, 0, (0, 0)), (16320, 8, 0, (2, 16))):
        assert exact_column_parts(_family_desc(I, 4096, v, 65536, kr), I) == want, (I, v, kr)
    d = _family_desc(28672, 4096, 8, 65536, 256)
    p = part_desc(d, 14336, 28672)
    assert (p.in_features, p.group_size, p.out_features, p.num_indices) == (14336, 14336, d.out_features, d.num_indices)
    assert p.weight_scale == d.weight_scale + 2 * 14336 and p.weight_bias == d.weight_bias + 2 * 14336 and p.centroids == d.centroids
    assert p.indices == d.indices and p.bias == d.bias and not p.perm


def test_one_pass_plan_for_two_and_three_tokens_without_gpu():
    """vptq_quant_gemv_sliced_tokens_one_pass: in how many window parts 2 / 3 tokens take one pass of the one-token kernel in the reference's
    roundings - 1 where the slice leaves room for (2 tokens + 4) bytes of every column, 2 / 4 where for half / a quarter of them (v = 8, one
    table), 0 else (column phases or the gather kernel); the same question per column part; the flags it answers for - host logic"""
    import ctypes as C
    from vptq_amd.utils.sliced import part_desc
    lib = B.lib()
    EX = B.GEMV_EXACT
    one = lib.vptq_quant_gemv_sliced_tokens_one_pass

    def want(I, v, kr, tokens):      # the LDS budget restated
        if tokens > (3 if v == 8 else 2) or (kr not in (0, 256) and v != 8) or (v == 16 and kr):
            return 0
        slices = lib.vptq_sliced_layout_supported_for(_family_desc(I, 4096, v, 65536, kr), EX)
        if not slices:
            return 0
        tab = 65536 // slices * v * 2
        res = 4096 if (v == 8 and kr == 256) else 0
        if tab + (I + 64) * (4 + 2 * tokens) + 64 + res <= 163840:
            return 1
        if v != 8 or kr not in (0, 256):
            return 0
        wc = (I + 31) // 32 * 8
        for parts in (2, 4):
            if tab + (min(4 // parts * wc, I) + 64) * (4 + 2 * tokens) + 64 + res <= 163840:
                return parts
        return 0
    for (I, v, kr) in ((2048, 8, 256), (4096, 8, 0), (4096, 8, 256), (5376, 8, 0), (8192, 8, 256), (12288, 8, 0), (14336, 8, 256), (16288, 8, 0),
                       (8192, 16, 0), (14336, 16, 0), (8192, 8, 65536), (14336, 8, 65536), (4096, 8, 65536), (8192, 8, 4096), (8192, 16, 65536)):
        d = _family_desc(I, 4096, v, 65536, kr)
        got = [one(d, t, EX) for t in (1, 2, 3, 4)]
        assert got == [0] + [want(I, v, kr, t) for t in (2, 3, 4)], (I, v, kr, got)
        assert one(d, 2, 0) == 0 and one(d, 2, EX | B.GEMV_FORCE_GENERIC) == 0        # (the folded form: column phases only)
    assert [one(_family_desc(4096, 4096, 8, 65536, 256), t, EX) for t in (2, 3)] == [2, 2]
    assert [one(_family_desc(14336, 4096, 8, 65536, 0), t, EX) for t in (2, 3)] == [2, 2]
    assert [one(_family_desc(8192, 4096, 8, 65536, 256), t, EX) for t in (2, 3)] == [1, 1]
    # window parts need the layout's window table, the whole-column pass does not
    buf = (C.c_char * 64)()
    p = C.addressof(buf)
    supf = lib.vptq_quant_gemv_sliced_tokens_supported_for

    def layout(dd, wstart):
        return (B.SlicedLayout * 1)(B.SlicedLayout(p, p, p, p, 2, 1, lib.vptq_sliced_layout_supported_for(dd, EX), 0, p if wstart else None))
    d4, d8 = _family_desc(4096, 4096, 8, 65536, 0), _family_desc(8192, 4096, 8, 65536, 0)
    assert supf(d4, layout(d4, True), 2, EX) == 1 and supf(d4, layout(d4, False), 2, EX) == 0
    assert supf(d8, layout(d8, True), 2, EX) == 1 and supf(d8, layout(d8, False), 2, EX) == 1
    # a 28672-column layer: not in one piece; each of its two column parts in two window parts
    dw = _family_desc(28672, 4096, 8, 65536, 256)
    assert [one(dw, t, EX) for t in (2, 3)] == [0, 0]
    pd = part_desc(dw, 0, 14336)
    assert [one(pd, t, EX) for t in (2, 3, 4)] == [2, 2, 0] and supf(pd, layout(pd, True), 3, EX) == 1


def test_exact_slice_rule_opt_in_for_two_tokens():
    """VPTQ_SLICED_SLICES=room2 (read once per process: a subprocess): the exact layouts of v = 8 layers take the smaller slice count only
    where TWO tokens' operands fit beside the slice - 4096-column layers then have 16 slices and their 2 / 3 tokens one pass"""
    import os
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "from vptq_amd import _backend as B\nfrom test_host_cpu import _family_desc\n"
            "f = B.lib().vptq_sliced_layout_supported_for\n"
            "print([f(_family_desc(I, 4096, v, 65536, kr), B.GEMV_EXACT) for (I, v, kr) in "
            "((4096, 8, 0), (4024, 8, 0), (4032, 8, 0), (3512, 8, 256), (3520, 8, 256), (8192, 8, 256), (4096, 16, 0), (5376, 16, 0))],"
            " [f(_family_desc(4096, 4096, 8, 65536, 256), 0)])") % (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden"))
    env = dict(os.environ, VPTQ_SLICED_SLICES="room2", VPTQ_TUNING="1")   # (tuning knobs are read only with VPTQ_TUNING=1)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.strip().splitlines()[-1] == "[16, 8, 16, 8, 16, 16, 16, 16] [8]", out.stdout
    env.pop("VPTQ_SLICED_SLICES")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().splitlines()[-1] == "[8, 8, 8, 8, 8, 16, 16, 16] [8]", out.stdout + out.stderr[-1000:]


def test_sliced_tokens_plan_without_gpu():
    """2 - 4 tokens over the sliced layouts (gemv_sliced_tok.hip): which layers the library takes - the activations of the
    tokens must fit the LDS beside the slice in at most 4 column phases - and the workspace it asks for: host logic"""
    import ctypes as C
    lib = B.lib()
    sup, wsb = lib.vptq_quant_gemv_sliced_tokens_supported, lib.vptq_quant_gemv_sliced_tokens_workspace_bytes
    buf = (C.c_char * 64)()   # (never dereferenced here)
    p = C.addressof(buf)

    def layouts(d, wstart=True):
        n = lib.vptq_sliced_layout_tables(d)
        return (B.SlicedLayout * n)(*[B.SlicedLayout(p, p, p, p, 2, 1, lib.vptq_sliced_layout_supported(d),
                                                     lib.vptq_sliced_layout_whole_table(d, t), p if wstart else None) for t in range(n)])
    d = _family_desc(8192, 8192, 8, 65536, 0)
    assert [sup(d, layouts(d), t) for t in (1, 2, 3, 4, 5)] == [0, 1, 1, 1, 0]
    assert sup(d, layouts(d, wstart=False), 2) == 0                       # layouts without the column windows' table: one token only
    # partial sums [tokens][slices x tables][N x v] floats (256-byte multiple) + the arrival counters (one per 16 rows)
    # (ONE size for both arithmetics: the exact layout of an 8192-column layer has 16 slices; one-table formats: + 3 x N x v
    # 64-bit accumulator words of the one-pass route for 2 / 3 exact tokens, behind the counters)
    acc = 3 * 8192 * 8
    assert wsb(d, 2) == 2 * 16 * 8192 * 4 + 256 + acc and wsb(d, 4) == 4 * 16 * 8192 * 4 + 256 + acc and wsb(d, 9) == 0 and wsb(d, 1) == 0
    d4 = _family_desc(4096, 4096, 8, 65536, 256)      # 5 - 8 tokens: 16 bytes of activations per column must fit in 4 phases
    assert [sup(d4, layouts(d4), t) for t in (5, 8, 9)] == [1, 1, 0] and wsb(d4, 8) == 8 * 8 * 4096 * 4 + 256 + 3 * 4096 * 8
    d2 = _family_desc(8192, 8192, 8, 65536, 65536)
    assert sup(d2, layouts(d2), 4) == 1 and wsb(d2, 3) == 3 * 16 * 8192 * 4 + 256 + acc   # (two tables of v = 8 have the one-pass route too)
    # ABI 9: the reference's roundings (one-table formats, an exact layout: its slice count), 8 + 2 x token slots bytes per column and phase
    EX = B.GEMV_EXACT
    supf = lib.vptq_quant_gemv_sliced_tokens_supported_for

    def exact_layout(dd, wstart=True):
        return (B.SlicedLayout * 1)(B.SlicedLayout(p, p, p, p, 2, 1, lib.vptq_sliced_layout_supported_for(dd, EX), 0, p if wstart else None))
    for (I, v, kr, want) in ((4096, 8, 256, [0, 1, 1, 1, 1, 0]), (8192, 8, 256, [0, 1, 1, 1, 1, 0]), (14336, 8, 0, [0, 1, 1, 1, 1, 0]),
                             (5376, 8, 0, [0, 1, 1, 0, 0, 0]), (16288, 8, 0, [0, 1, 1, 0, 0, 0]), (8192, 16, 0, [0, 1, 1, 1, 1, 0]),
                             (8192, 8, 65536, [0, 1, 0, 0, 0, 0]), (8192, 16, 65536, [0] * 6), (8192, 8, 4096, [0, 1, 0, 0, 0, 0]), (14336, 8, 65536, [0] * 6)):   # (two tables, v = 8: 2 / 3 tokens in one pass where they fit)
        dd = _family_desc(I, 4096, v, 65536, kr)
        assert [supf(dd, exact_layout(dd), t, EX) for t in (1, 2, 4, 5, 8, 9)] == want, (I, v, kr)
        assert supf(dd, exact_layout(dd), 2, EX | B.GEMV_FORCE_GENERIC) == 0
    # 2 / 3 tokens of a layer whose exact layout has 16 slices (+ 2 tokens + 4 bytes per column fit beside the 64 KiB slice): ONE pass
    # of the one-token kernel - no column windows needed; 4 tokens and narrow layers (8 slices of 128 KiB): the column-phase kernel
    assert [supf(d, exact_layout(d, wstart=False), t, EX) for t in (2, 3, 4)] == [1, 1, 0]
    assert [supf(d4, exact_layout(d4, wstart=False), t, EX) for t in (2, 3, 4)] == [0, 0, 0]
    d14 = _family_desc(14336, 4096, 8, 65536, 256)
    assert [supf(d14, exact_layout(d14, wstart=False), t, EX) for t in (2, 3)] == [0, 0] and supf(d14, exact_layout(d14), 2, EX) == 1
    d16 = _family_desc(8192, 4096, 16, 65536, 0)
    assert [supf(d16, exact_layout(d16, wstart=False), t, EX) for t in (2, 3)] == [1, 0]
    assert supf(d, layouts(d), 2, EX) == 0                 # the folded form's 8 slices: not the exact layout of an 8192-column layer
    assert supf(d4, layouts(d4), 8, EX) == 1               # 4096 columns: the same 8 slices, the same layout
    assert supf(d, layouts(d), 4, 0) == sup(d, layouts(d), 4) == 1
    assert sup(_family_desc(28672, 8192, 8, 65536, 0), layouts(_family_desc(28672, 8192, 8, 65536, 0)), 4) == 1   # 16 slices of 64 KiB
    dv = _family_desc(8192, 8192, 16, 65536, 65536)
    assert sup(dv, layouts(dv), 2) == 1
    assert sup(_family_desc(8192, 8192, 12, 65536, 0), layouts(d), 2) == 0   # not a layer of the sliced path at all


def test_one_launch_rule_for_two_to_four_tokens():
    """VQuantLinear._sliced_one_launch: which layers send 2 - 4 tokens through the token kernel over the sliced layouts (the
    measured table of profiles/r04/sliced_tokens_one_launch.txt as a rule) - host logic, no GPU"""
    import types
    import torch
    from vptq_amd.layers.vqlinear import VQuantLinear

    def layer(I, O, v, kr):
        return types.SimpleNamespace(indices=torch.empty(1, O // v, 1), group_size=I, vector_len=v, out_features=O,
                                     num_res_centroids=kr, enable_residual=kr > 0)

    def rule(I, O, v, kr, tokens, supported=True, exact=False, slices=8, one_pass=False):
        class SL:       # (stands in for vptq_amd.utils.sliced.SlicedGemv: what the library answers + the per-layout cache)
            def tokens_supported(self, t):
                return supported
        sl = SL()
        sl.exact, sl.slices = exact, slices
        sl.tokens_one_pass = lambda t: bool(one_pass)
        sl.tokens_window_parts = lambda t: int(one_pass)
        return VQuantLinear._sliced_one_launch(layer(I, O, v, kr), sl, tokens)
    for t in (2, 3, 4):
        assert rule(8192, 8192, 8, 0, t) and rule(4096, 4096, 8, 256, t) and rule(4096, 14336, 8, 65536, t) and rule(8192, 8192, 16, 0, t)
        assert not rule(2048, 512, 8, 256, t)                     # tiny: launch-bound on every route
        assert not rule(8192, 8192, 16, 1024, t)                  # v = 16 with a small residual table: the gather kernel holds it in LDS
        assert not rule(8192, 8192, 8, 0, t, supported=False)     # the library does not take it (no wstart, no room in LDS)
    assert rule(4096, 14336, 16, 65536, 2) and not rule(4096, 14336, 16, 65536, 4) and rule(8192, 8192, 16, 65536, 4)
    # the reference's roundings (profiles/r05/sliced_tokens_exact.txt): large v = 8 layers whose exact layout has 16 slices
    assert not rule(8192, 8192, 8, 256, 2, exact=True, slices=16)      # 2 tokens: two launches of the one-token kernel instead
    for t in (3, 4):
        assert rule(8192, 8192, 8, 256, t, exact=True, slices=16) and rule(14336, 4096, 8, 0, t, exact=True, slices=16)
        assert not rule(4096, 14336, 8, 256, t, exact=True, slices=8) and not rule(4096, 4096, 8, 0, t, exact=True, slices=8)
        assert not rule(8192, 8192, 16, 0, t, exact=True, slices=32) and not rule(8192, 2048, 8, 0, t, exact=True, slices=16)
        assert not rule(8192, 8192, 8, 256, t, exact=True, slices=16, supported=False)
    # ... and 2 / 3 tokens in ONE PASS of the one-token kernel wherever the library takes that (16-slice layouts, any size)
    for t in (2, 3):
        assert rule(8192, 8192, 8, 256, t, exact=True, slices=16, one_pass=True) and rule(8192, 1024, 8, 0, t, exact=True, slices=16, one_pass=True)
        assert not rule(2048, 8192, 8, 0, t, exact=True, slices=8, one_pass=True) and not rule(8192, 8192, 16, 0, t, exact=True, slices=32, one_pass=True)
        # two tables of v = 8: large layers with >= 4096 residual centroids (the one-token rule's sizes)
        assert rule(8192, 8192, 8, 65536, t, exact=True, slices=16, one_pass=True) and rule(8192, 8192, 8, 4096, t, exact=True, slices=16, one_pass=True)
        assert not rule(8192, 1024, 8, 65536, t, exact=True, slices=16, one_pass=True) and not rule(8192, 8192, 8, 1024, t, exact=True, slices=16, one_pass=True)
        assert not rule(8192, 8192, 8, 65536, t, exact=True, slices=16, one_pass=False)
    assert not rule(8192, 8192, 8, 65536, 4, exact=True, slices=16, one_pass=True)
    # ... and in WINDOW PARTS (the library's answer 2 / 4: half / a quarter of the columns staged per workgroup) on layers of >= 6 M index elements
    for t in (2, 3):
        assert rule(14336, 4096, 8, 256, t, exact=True, slices=16, one_pass=2) and rule(4096, 14336, 8, 0, t, exact=True, slices=8, one_pass=2)
        assert not rule(4096, 4096, 8, 0, t, exact=True, slices=8, one_pass=2) and not rule(14336, 512, 8, 0, t, exact=True, slices=16, one_pass=2)
