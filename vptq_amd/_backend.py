"""ctypes binding of libvptq_hip.so (C ABI: include/vptq_hip.h).

This is the only place the package talks to native code.  There is NO CPU or
pure-torch fallback: if the library is missing, or a tensor is not on a ROCm
device, the ops raise (the reference instead prints a warning and silently runs
"extremely slow" torch code, vptq/ops/quant_gemm.py:28-40).
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# VPTQ_HIP_LIB: load another build of the library (A/B runs of tools/)
LIB_PATH = os.environ.get("VPTQ_HIP_LIB") or os.path.join(_HERE, "libvptq_hip.so")


def tune_env(name: str, default=None):
    """Tuning / A-B knobs (tools/README.md lists them) are read only when VPTQ_TUNING=1 is set - here and in the library
    (csrc/common.h:tune_env): the package's behaviour does not depend on stray environment variables.  The four PRODUCT
    knobs are read unconditionally: VPTQ_ARITHMETIC (reference | selective | folded), VPTQ_SLICED_LAYOUT (auto | 0 | 1),
    VPTQ_FUSED_GEMM_MAX_TOKENS, VPTQ_HIP_LIB (another build of the library)."""
    return os.environ.get(name, default) if os.environ.get("VPTQ_TUNING") == "1" else default

ABI_VERSION = 10
DTYPE_F16, DTYPE_BF16 = 0, 1
GEMV_FAST_MATH = 1 << 0
GEMV_FORCE_GENERIC = 1 << 1
GEMV_EXACT = 1 << 2
GEMV_FORCE_MFMA = 1 << 3
GEMV_FORCE_VALU = 1 << 4
GEMV_OUT_F32 = 1 << 5     # y is float32: un-rounded sums (row-parallel partial outputs)
GEMV_MAX_TOKENS = 64       # most vptq_quant_gemv accepts (any layer: 16); per layer: vptq_quant_gemv_max_tokens
GEMV_ANY_FORMAT_TOKENS = 8  # the fused GEMV is the faster path for every format up to here
GEMV_CHAIN_DEPENDENT = 1 << 6  # vptq_quant_gemv_chain: layer i + 1 reads what layer i wrote
GEMV_FORCE_BATCHED = 1 << 7    # the one-pass batched-decode kernel wherever eligible (tests, A/B)
GEMV_COLUMN_PARTS = 1 << 8     # vptq_quant_gemv_sliced_grouped: the descriptors are column ranges of ONE layer (shared y / workspace)
GEMV_SELECTIVE = 1 << 9        # folded form + the reference's roundings on the activation columns that dominate the token (ABI 10)
# return codes (include/vptq_hip.h)
E_NULL, E_SHAPE, E_UNSUPPORTED, E_ALIGN, E_TOKENS, E_WORKSPACE = -1, -2, -3, -4, -5, -6
GROUP_MAX = 64
CHAIN_MAX = 1024

_vp = C.c_void_p


class SlicedLayout(C.Structure):
    """Mirror of `VptqSlicedLayout` (include/vptq_hip.h)."""
    _fields_ = [("elems", C.c_void_p), ("blocks", C.c_void_p), ("first", C.c_void_p), ("res", C.c_void_p),
                ("rows_per_wave", C.c_int32), ("elems_per_lane", C.c_int32), ("n_slices", C.c_int32), ("whole_table", C.c_int32),
                ("wstart", C.c_void_p)]


class LayerDesc(C.Structure):
    """Mirror of `VptqLayerDesc` (include/vptq_hip.h)."""
    _fields_ = [(n, C.c_int32) for n in (
        "in_features", "out_features", "vector_len", "num_codebooks", "group_size",
        "num_centroids", "num_res_centroids", "index_bits", "res_bits", "row_words",
        "num_indices", "outlier_size", "outlier_vector_len", "num_outlier_centroids",
        "num_outlier_indices", "dtype")] + [(n, _vp) for n in (
            "indices", "centroids", "res_centroids", "outlier_indices", "outlier_centroids",
            "perm", "inv_perm", "weight_scale", "weight_bias", "bias", "scale_permuted",
            "bias_permuted", "prefetch")] + [("prefetch_bytes", C.c_int64)]


class V2Desc(C.Structure):
    """Mirror of `VptqV2Desc`."""
    _fields_ = [(n, C.c_int32) for n in (
        "in_features", "out_features", "vector_len", "num_centroids", "num_res_centroids",
        "res_index_bytes", "dtype", "reserved")] + [(n, _vp) for n in (
            "indices", "centroids", "res_indices", "res_centroids", "scale_weights",
            "scale_bias", "bias")]


# every symbol include/vptq_hip.h declares: (restype, argtypes)
EXPORTS = {
    "vptq_abi_version": (C.c_int, []),
    "vptq_last_error": (C.c_char_p, []),
    "vptq_quant_gemv": (C.c_int, [C.POINTER(LayerDesc), _vp, _vp, C.c_int, C.c_int, _vp,
                                  C.c_size_t, _vp]),
    "vptq_quant_gemv_workspace_bytes": (C.c_size_t, [C.POINTER(LayerDesc), C.c_int, C.c_int]),
    "vptq_quant_gemv_max_tokens": (C.c_int, [C.POINTER(LayerDesc)]),
    "vptq_quant_gemv_grouped": (C.c_int, [C.POINTER(LayerDesc), C.c_int, C.POINTER(_vp),
                                          C.POINTER(_vp), C.c_int, C.c_int, _vp]),
    "vptq_quant_gemv_chain": (C.c_int, [C.POINTER(LayerDesc), C.c_int, C.POINTER(_vp), C.POINTER(_vp),
                                        C.c_int, C.c_int, _vp, C.c_size_t, _vp]),
    "vptq_quant_gemv_chain_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "vptq_quant_gemv_chain_workspace_bytes_for": (C.c_size_t, [C.POINTER(LayerDesc), C.c_int, C.c_int]),
    "vptq_quant_gemv_chain_kernel_name": (C.c_char_p, [C.POINTER(LayerDesc), C.c_int, C.c_int, C.c_int]),
    "vptq_dequant": (C.c_int, [C.POINTER(LayerDesc), _vp, _vp]),
    "vptq_sliced_layout_supported": (C.c_int, [C.POINTER(LayerDesc)]),
    "vptq_sliced_layout_tables": (C.c_int, [C.POINTER(LayerDesc)]),
    "vptq_sliced_layout_whole_table": (C.c_int, [C.POINTER(LayerDesc), C.c_int]),
    "vptq_quant_gemv_sliced_workspace_bytes": (C.c_size_t, [C.POINTER(LayerDesc)]),
    "vptq_quant_gemv_sliced_workspace_bytes_for": (C.c_size_t, [C.POINTER(LayerDesc), C.c_int]),
    "vptq_quant_gemv_sliced_selective_supported": (C.c_int, [C.POINTER(LayerDesc)]),
    "vptq_quant_gemv_sliced": (C.c_int, [C.POINTER(LayerDesc), C.POINTER(SlicedLayout), _vp, _vp, C.c_int, _vp,
                                         C.c_size_t, _vp]),
    "vptq_quant_gemv_sliced_tokens_supported": (C.c_int, [C.POINTER(LayerDesc), C.POINTER(SlicedLayout), C.c_int]),
    "vptq_quant_gemv_sliced_tokens_supported_for": (C.c_int, [C.POINTER(LayerDesc), C.POINTER(SlicedLayout), C.c_int, C.c_int]),
    "vptq_quant_gemv_sliced_tokens_one_pass": (C.c_int, [C.POINTER(LayerDesc), C.c_int, C.c_int]),
    "vptq_quant_gemv_sliced_tokens_workspace_bytes": (C.c_size_t, [C.POINTER(LayerDesc), C.c_int]),
    "vptq_quant_gemv_sliced_tokens": (C.c_int, [C.POINTER(LayerDesc), C.POINTER(SlicedLayout), _vp, _vp, C.c_int, C.c_int, _vp,
                                                C.c_size_t, _vp]),
    "vptq_quant_gemv_sliced_tokens_grouped": (C.c_int, [C.POINTER(LayerDesc), C.POINTER(SlicedLayout), C.c_int, _vp, C.POINTER(_vp), C.c_int, C.c_int,
                                                        C.POINTER(_vp), C.POINTER(C.c_size_t), _vp]),
    "vptq_quant_gemv_sliced_grouped": (C.c_int, [C.POINTER(LayerDesc), C.POINTER(SlicedLayout), C.c_int, _vp, C.POINTER(_vp), C.c_int,
                                                 C.POINTER(_vp), C.POINTER(C.c_size_t), _vp]),
    "vptq_quant_gemm_supported": (C.c_int, [C.POINTER(LayerDesc)]),
    "vptq_quant_gemm_workspace_bytes": (C.c_size_t, [C.POINTER(LayerDesc), C.c_int]),
    "vptq_quant_gemm": (C.c_int, [C.POINTER(LayerDesc), _vp, _vp, C.c_int, C.c_int, _vp, C.c_size_t, _vp]),
    "vptq_quant_gemv_v2": (C.c_int, [C.POINTER(V2Desc), _vp, _vp, C.c_int, C.c_int, _vp]),
    "vptq_quant_gemv_kernel_name": (C.c_char_p, [C.POINTER(LayerDesc), C.c_int, C.c_int]),
    "vptq_sliced_layout_supported_for": (C.c_int, [C.POINTER(LayerDesc), C.c_int]),
    "vptq_quant_gemv_grouped_kernel_name": (C.c_char_p, [C.POINTER(LayerDesc), C.c_int, C.c_int, C.c_int]),
}

_lib = None


class VptqBackendError(RuntimeError):
    pass


def lib():
    """Load libvptq_hip.so once; raise loudly if it is not there."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise VptqBackendError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` or `make -C vptq_amd/csrc`.  vptq_amd has no CPU / torch fallback.")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in EXPORTS.items():
            fn = getattr(l, name)  # AttributeError = ABI mismatch, also loud
            fn.restype, fn.argtypes = res, args
        if l.vptq_abi_version() != ABI_VERSION:
            raise VptqBackendError(
                f"ABI mismatch: library {l.vptq_abi_version()} vs binding {ABI_VERSION}")
        _lib = l
    return _lib


def is_available() -> bool:
    return os.path.exists(LIB_PATH)


def check(rc: int, what: str):
    if rc != 0:
        msg = lib().vptq_last_error().decode(errors="replace")
        raise RuntimeError(f"{what} failed (code {rc}): {msg}")


def dtype_code(dt: torch.dtype) -> int:
    if dt == torch.float16:
        return DTYPE_F16
    if dt == torch.bfloat16:
        return DTYPE_BF16
    raise TypeError(f"vptq_amd supports float16 / bfloat16 activations, got {dt}")


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def require_device(*tensors: Optional[torch.Tensor]):
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise VptqBackendError(
                "vptq_amd runs on ROCm devices only; got a tensor on "
                f"{t.device}.  There is no CPU fallback.")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError(f"tensors on different devices: {dev} vs {t.device}")
        if not t.is_contiguous():
            raise RuntimeError("vptq_amd needs contiguous tensors")
    return dev


_GEMV_WS = {}        # (device index, stream handle) -> uint8 tensor
_GEMV_WS_RETIRED = []   # outgrown buffers stay alive: a captured hipGraph may still point at them


def gemv_workspace(dev_index: int, stream_ptr: int, nbytes: int):
    """(pointer, bytes) of the scratch buffer `vptq_quant_gemv` may use on this (device, stream), or
    (None, 0).  One buffer per stream: calls on a stream are ordered, so they can share it.  Nothing is
    allocated while the stream is being captured into a graph (the buffer would belong to the graph's
    private pool): such a call runs without a workspace, i.e. with the kernels that need none."""
    if nbytes <= 0:
        return None, 0
    key = (dev_index, stream_ptr)
    t = _GEMV_WS.get(key)
    if t is None or t.numel() < nbytes:
        if torch.cuda.is_current_stream_capturing():
            return None, 0
        if t is not None:
            _GEMV_WS_RETIRED.append(t)
        t = torch.empty(max(int(nbytes), 2 << 20), dtype=torch.uint8, device=torch.device("cuda", dev_index))
        _GEMV_WS[key] = t
    return t.data_ptr(), t.numel()


def current_stream_ptr(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def tensor_version(t: Optional[torch.Tensor]) -> int:
    """`t._version`, or -1 for tensors that do not track one (created under
    `torch.inference_mode()`: reading `_version` raises there).  Such tensors can only be
    written in place inside inference mode; derived state is then keyed on the storage
    pointer alone."""
    if t is None:
        return 0
    if t.is_inference():
        return -1
    return t._version


def _derived(owner: torch.Tensor, name: str, key, build):
    """Derived state cached ON the owning tensor object (so it dies with it and is never
    confused with another tensor that later reuses the same address), re-built when the
    owner's storage pointer or version counter changes."""
    slot = getattr(owner, "_vptq_derived", None)
    if slot is None:
        slot = {}
        owner._vptq_derived = slot
    hit = slot.get(name)
    if hit is None or hit[0] != key:
        hit = (key, build())
        slot[name] = hit
    return hit[1]


# ---- load-time gate of the opt-in "folded" decode arithmetic -------------------------------------------------------
# The folded form  y = sum (c + r) * f16(s x) + sum b x  is as close to exact arithmetic as the reference CPU path
# is: the reference rounds every weight three times (vptq/ops/quant_gemm.py:121,155-156), which puts ITS un-rounded
# sums 2-7e-4 of max|y| away from exact math.  What the parity bar (1e-3 of max|y|) has room for:
# a rounded output differs from the reference's by n ulps where n <= |un-rounded distance| / ulp + 1, one fp16 ulp of
# the top binade is 2^-10 * 2^k with max|y| = m 2^k, 1 <= m < 2, so ONE flip costs 9.77e-4 / m <= 1e-3 and TWO exceed
# the bar while m < 1.95: a layer is safe while its un-rounded distance stays under one top-binade ulp, i.e. under
# 9.77e-4 / m of max|y|.
#
# Round 5: the gate MEASURES that distance on the layer's own tensors instead of predicting it from tensor statistics
# (rounds 3-4: rms(b) against rms(s) rms(c + r) and a count of distinct index rows - tuned on four synthetic
# distributions).  At descriptor build the layer runs both forms through the library - folded and
# VPTQ_GEMV_EXACT (the reference's three roundings: >= 99 % bit-identical, un-rounded distance ~1e-6), float32
# outputs - on PROBE activations: N(0, 1); N(0, 1) with 8 channels x 40 (outlier channels); |N(0, 1)| (a large mean:
# sum b x as large as it gets); N(0, 1) projected orthogonal to the weight bias (sum b x cancels: what exposes a
# bias-dominated layer).  The layer keeps the folded form while max|y_folded - y_exact| / max|y_exact| stays under
# FOLDED_MAX_PROBE_DISTANCE on every probe.  Bias-dominated layers (1.6e-3 at |b| = 8 |w s|) and layers with few distinct
# outputs (the reference test's cyclic index pattern: max|y| is a maximum over 8 values, not thousands) land far
# above the line; random layers of the LLM-like and the reference-test distributions at 2-7e-4 (tools/gate_probe_study.py,
# profiles/r05/gate_probe_study_*.txt).  One device -> host read per descriptor build, i.e. per layer load.
# ---- which arithmetic the decode GEMV evaluates by default ----------------------------------------------------------
# "reference" (the default since round 5): every weight is rebuilt with the reference CPU path's three 16-bit roundings
# (VPTQ_GEMV_EXACT: w = f16(f16(f16(c + r) * s) + b), vptq/ops/quant_gemm.py:121,155-156; un-rounded sums ~1e-6 of max|y|
# from the reference's, >= 99 % of the outputs bit-identical, the rest one flip of the last bit) - the form SURVEY 7.2(5)
# specified as the default and the only one that stays inside the 1e-3 bar for EVERY activation.
# "folded" (opt-in: VPTQ_ARITHMETIC=folded, set_arithmetic("folded")): y = sum (c + r) f16(s x) + sum b x in fp32 - 25-35 %
# faster on the canonical format and the only arithmetic of the sliced layouts of the large-codebook formats.  It is as
# close to exact math as the reference is, but it does not repeat the reference's rounding errors: with dense
# activations those average out over the columns (0 of 12 288 layers above the bar, profiles/r04/fuzz_count_*), with
# activations dominated by a few channels - massive-activation channels, 90 % sparse inputs - they do not: 3 of 1000
# checkpoint-like fp16 layers at 1.05 - 1.26e-3 and 1 of 500 bf16 layers at 9.8e-3 against bars of 1e-3 / 8e-3
# (tools/gpu_gate_count.py, profiles/r05/gate_count_*_folded_default.txt).  No load-time gate can see that coming - it
# depends on the activation - so the bar-safe form is the default and the fast one is the user's decision.
# "selective" (opt-in, round 6: VPTQ_ARITHMETIC=selective, set_arithmetic("selective")): the reference's roundings wherever an
# activation column dominates the token (|f16(s x)| >= 6 x the rms over the layer's columns: blocks of 128 columns that hold such
# a column are rebuilt bit-exactly), the folded form on the rest - VPTQ_GEMV_SELECTIVE, where a kernel implements it (the
# persistent chain launch, fp16); every other route takes the reference's roundings, as do the layers the load-time gate below
# refuses.  It removes the folded form's activation-dependent failures (chain route, same ten families x five activation
# kinds: 30 of 4100 layers above the bar folded, 2 of 12 300 selective at 1.00e-3 / 1.09e-3 - both dense-activation tail
# events no run-time test can see; profiles/r06/count_chain_*.txt) at 95 % of its speed, and is still not bit-equivalent:
# ~55 - 65 % of the outputs bit-identical, worst ~9.8e-4 = one flip of the last bit at the top binade.
_ARITH = {"folded": os.environ.get("VPTQ_ARITHMETIC", "reference").strip().lower() in ("folded", "fast") or
          tune_env("VPTQ_FOLDED", "0") == "1",
          "selective": os.environ.get("VPTQ_ARITHMETIC", "reference").strip().lower() == "selective", "generation": 0}


def set_arithmetic(mode: str) -> None:
    """"reference" (default), "selective" or "folded"; layers rebuild their descriptors (and drop or build their sliced
    layouts) at their next call.  Not inside a stream capture."""
    mode = mode.strip().lower()
    if mode not in ("reference", "exact", "selective", "folded", "fast"):
        raise ValueError("arithmetic is 'reference', 'selective' or 'folded'")
    _ARITH["folded"] = mode in ("folded", "fast")
    _ARITH["selective"] = mode == "selective"
    _ARITH["generation"] += 1


def arithmetic() -> str:
    return "folded" if _ARITH["folded"] else ("selective" if _ARITH["selective"] else "reference")


def layer_arithmetic_flags(gate_passed: bool) -> int:
    """flags of a layer's launches in the process's arithmetic: VPTQ_GEMV_EXACT (reference; any layer the load-time gate refuses),
    VPTQ_GEMV_SELECTIVE (selective) or 0 (folded)"""
    if not gate_passed:
        return GEMV_EXACT
    return 0 if _ARITH["folded"] else (GEMV_SELECTIVE if _ARITH["selective"] else GEMV_EXACT)


def arithmetic_generation() -> int:
    return _ARITH["generation"]


FOLDED_MAX_PROBE_DISTANCE = {torch.float16: 7.5e-4, torch.bfloat16: 6.0e-3}   # (bf16: 8 mantissa bits fewer... 3: x 8)
# the selective arithmetic's gate sits lower: what is left of its failures are LAYER properties (a bias-dominated layer at a probe
# distance of 7.3e-4 went 1.09e-3 above the bar on a dense activation, profiles/r06/count_chain_selective_seed0_probe.txt); 6.5e-4
# sends 4 - 9 % of checkpoint-like layers (a third of the un-centred "big-bias" family) to the reference's roundings
SELECTIVE_MAX_PROBE_DISTANCE = {torch.float16: 6.5e-4, torch.bfloat16: 5.2e-3}   # (bf16: x 8, as above)
FOLDED_MIN_DISTINCT_ROWS = 32   # (kept as a cheap pre-filter: tiny layers and repeating index rows never take the folded form)
_ROW_SAMPLE = 64
_PROBE_SEED = 0x5eed


def distinct_index_rows(indices: torch.Tensor) -> int:
    """lower bound of the number of distinct packed index rows: min(rows, distinct among 64 evenly spaced rows)"""
    rows = indices.reshape(-1, indices.shape[-1])[: indices.shape[-2]]   # (first codebook group)
    n = rows.shape[0]
    if n <= 1:
        return n
    pick = torch.linspace(0, n - 1, min(n, _ROW_SAMPLE), device=rows.device).round().long()
    r = rows[pick].to(torch.int64)
    w = torch.arange(1, r.shape[1] + 1, device=r.device, dtype=torch.int64) * 0x9E3779B1 + 1
    h = (r * w).sum(1)   # (int64 arithmetic wraps: a position-weighted checksum per row)
    return int(torch.unique(h).numel())


def probe_activations(in_features: int, weight_bias: Optional[torch.Tensor], dtype, dev) -> torch.Tensor:
    """[4, in_features] probe activations of the folded-form gate (deterministic: the same layer gets the same answer)"""
    g = torch.Generator(device=dev).manual_seed(_PROBE_SEED + in_features)
    r = torch.randn(4, in_features, generator=g, device=dev, dtype=torch.float32)
    hot = torch.randint(0, in_features, (8,), generator=g, device=dev)
    r[1, hot] *= 40.0
    r[2] = r[2].abs()
    if weight_bias is not None:
        b = weight_bias.detach().float().reshape(-1)
        bb = (b * b).sum()
        r[3] = torch.where(bb > 0, r[3] - ((r[3] * b).sum() / bb.clamp_min(1e-30)) * b, r[3])
    return r.to(dtype).contiguous()


def folded_probe_distance(desc, in_features: int, out_features: int, weight_bias, dtype, dev, folded=None) -> torch.Tensor:
    """max over the probes of max|y_folded - y_exact| / max|y_exact| on float32 outputs (a 0-dim device tensor).
    `folded(x_row, y_f32) -> bool`: another folded route than the library's default one (the sliced layouts)."""
    fn = lib().vptq_quant_gemv
    xs = probe_activations(in_features, weight_bias, dtype, dev)
    ya = torch.empty(xs.shape[0], out_features, dtype=torch.float32, device=dev)
    yb = torch.empty_like(ya)
    with torch.cuda.device(dev):
        sp = current_stream_ptr(dev)
        for i in range(xs.shape[0]):
            if folded is not None:
                if not folded(xs[i], ya[i]):
                    return torch.full((), float("inf"), device=dev)
            else:
                check(fn(desc, xs[i].data_ptr(), ya[i].data_ptr(), 1, GEMV_OUT_F32, None, 0, sp), "vptq_quant_gemv")
            check(fn(desc, xs[i].data_ptr(), yb[i].data_ptr(), 1, GEMV_OUT_F32 | GEMV_EXACT, None, 0, sp), "vptq_quant_gemv")
    den = yb.abs().amax(1).clamp_min(1e-30)
    d = ((ya - yb).abs().amax(1) / den)
    # (a probe whose exact output is not finite says nothing about the arithmetic: inf / nan weights are the caller's)
    d = torch.where(torch.isfinite(d), d, torch.zeros_like(d))
    return d.amax()


def folded_form_is_safe(indices, centroids, res_centroids, weight_scale, weight_bias, desc=None,
                        in_features: int = 0, out_features: int = 0) -> bool:
    """One device -> host read per call: call it once per set of tensors (descriptor build / functional-API cache).
    With `desc` (the layer's descriptor): the measured gate; without: only the pre-filter."""
    if not (_ARITH["folded"] or _ARITH["selective"]):
        return False   # the default: the reference's roundings for EVERY layer (layers without scale / bias too: f16(c + r) is a rounding)
    if weight_scale is None or weight_bias is None or not weight_scale.is_cuda:
        return True
    with torch.no_grad():
        if indices is not None and indices.dim() == 3:
            if indices.shape[1] < FOLDED_MIN_DISTINCT_ROWS or distinct_index_rows(indices) < FOLDED_MIN_DISTINCT_ROWS:
                return False
        if desc is None or torch.cuda.is_current_stream_capturing():
            return desc is None   # (no launch + read-back inside a capture: the reference's roundings serve until a rebuild)
        lim = (SELECTIVE_MAX_PROBE_DISTANCE if _ARITH["selective"] else FOLDED_MAX_PROBE_DISTANCE).get(centroids.dtype)
        if lim is None and _ARITH["selective"]:
            return False   # (selective roundings exist for fp16: every other dtype takes the reference's)
        if lim is None:
            return True
        name = lib().vptq_quant_gemv_kernel_name(desc, 1, 0)
        name_x = lib().vptq_quant_gemv_kernel_name(desc, 1, GEMV_EXACT)
        if name is not None and name == name_x:
            return True   # (one kernel either way: this layer's default route IS the reference's roundings - nothing to gate)
        try:
            d = folded_probe_distance(desc, in_features, out_features, weight_bias, centroids.dtype, weight_scale.device)
        except VptqBackendError:
            return True   # (a layer the library refuses shows that at its first real call, with the library's message)
        return bool((d <= lim).item())


def inverse_perm(perm: torch.Tensor) -> torch.Tensor:
    """argsort(perm) as int16 (uint16 bit pattern), cached per tensor version.

    The reference recomputes this with a sort kernel on EVERY forward
    (vptq/ops/quant_gemm.py:208-211)."""
    def build():
        p = perm.detach().view(torch.int16).to(torch.int64) & 0xFFFF
        inv = torch.argsort(p)
        # store the uint16 bit pattern in an int16 tensor
        return torch.where(inv >= 32768, inv - 65536, inv).to(torch.int16)
    return _derived(perm, "inv_perm", (perm.data_ptr(), tensor_version(perm), perm.numel()), build)


def permuted_norm(perm: torch.Tensor, t: torch.Tensor, name: str) -> torch.Tensor:
    """t[perm[c]] in column order (derived state): lets the GEMV read scale / bias with
    coalesced loads when a permutation is present."""
    def build():
        idx = perm.detach().view(torch.int16).to(torch.int64) & 0xFFFF
        return t.detach()[idx].contiguous()
    key = (perm.data_ptr(), tensor_version(perm), t.data_ptr(), tensor_version(t), t.numel())
    return _derived(t, "permuted_" + name, key, build)


def make_layer_desc(*, indices, centroids, res_centroids, outlier_indices, outlier_centroids,
                    perm, weight_scale, weight_bias, bias, in_features, out_features,
                    vector_len, num_codebooks, num_centroids, num_res_centroids, group_size,
                    outlier_size, outlier_vector_len, num_outlier_centroids,
                    need_inv_perm=False, prefetch=None):
    """Build a LayerDesc from reference-format tensors.  Returns (desc, keepalive)."""
    if indices.dtype != torch.int32:
        # reference: TORCH_CHECK_EQ(q_indice.dtype(), torch::kInt) (csrc/quant_gemv.cu:258)
        raise RuntimeError("`indices` must be packed int32 (is_indice_packed=True)")
    d = LayerDesc()
    d.in_features, d.out_features = in_features, out_features
    d.vector_len, d.num_codebooks, d.group_size = vector_len, num_codebooks, group_size
    d.num_centroids = num_centroids
    kr = num_res_centroids if (res_centroids is not None and num_res_centroids > 0) else 0
    d.num_res_centroids = kr
    d.index_bits = int(math.ceil(math.log2(num_centroids)))
    d.res_bits = int(math.ceil(math.log2(kr))) if kr > 0 else 0
    d.row_words = indices.shape[-1]
    d.num_indices = (out_features + vector_len - 1) // vector_len
    has_out = outlier_centroids is not None and outlier_size > 0
    d.outlier_size = outlier_size if has_out else 0
    d.outlier_vector_len = outlier_vector_len if has_out else 0
    d.num_outlier_centroids = num_outlier_centroids if has_out else 0
    d.num_outlier_indices = ((out_features + outlier_vector_len - 1) // outlier_vector_len
                             if has_out else 0)
    d.dtype = dtype_code(centroids.dtype)
    keep = [indices, centroids, res_centroids, outlier_indices, outlier_centroids, perm,
            weight_scale, weight_bias, bias]
    d.indices, d.centroids = _ptr(indices), _ptr(centroids)
    d.res_centroids = _ptr(res_centroids) if kr > 0 else None
    d.outlier_indices = _ptr(outlier_indices) if has_out else None
    d.outlier_centroids = _ptr(outlier_centroids) if has_out else None
    d.perm = _ptr(perm)
    d.inv_perm = None
    if perm is not None and need_inv_perm:
        inv = inverse_perm(perm)
        keep.append(inv)
        d.inv_perm = _ptr(inv)
    norm = weight_scale is not None and weight_bias is not None
    d.weight_scale = _ptr(weight_scale) if norm else None
    d.weight_bias = _ptr(weight_bias) if norm else None
    d.bias = _ptr(bias)
    d.scale_permuted = d.bias_permuted = None
    if perm is not None and norm and not need_inv_perm:
        sp = permuted_norm(perm, weight_scale, "scale")
        bp = permuted_norm(perm, weight_bias, "bias")
        keep += [sp, bp]
        d.scale_permuted, d.bias_permuted = _ptr(sp), _ptr(bp)
    d.prefetch, d.prefetch_bytes = None, 0
    if prefetch is not None:
        keep.append(prefetch)
        d.prefetch, d.prefetch_bytes = prefetch.data_ptr(), prefetch.numel() * prefetch.element_size()
    return d, keep
