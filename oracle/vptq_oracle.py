"""CPU ORACLE (numpy) for VPTQ's fused dequant+GEMV hot path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it.  Nothing under ``vptq_amd/`` imports it; the product path fails loudly when
the HIP extension is missing instead of falling back to this.

It is a restatement (not a copy) of the reference's pure-torch CPU path:

* bit-stream unpack ........ /root/reference/vptq/utils/pack.py:105-139
* bit-stream pack .......... /root/reference/vptq/utils/pack.py:26-67
* dequant to dense W[O,I] .. /root/reference/vptq/ops/quant_gemm.py:43-158
* y = F.linear(x, W, bias) . /root/reference/vptq/ops/quant_gemm.py:274
* v2 wire format ........... /root/reference/tests/test_quant_gemv.py:49-109

Parity pin: ``tests/golden/*.npz`` were produced by importing the real Python
reference in the build container (``tests/golden/gen_golden.py``); the oracle
is checked bit-exactly (W) / to a few ulp (y) against them in
``tests/test_oracle_golden.py``.  When ``/root/reference`` is present the same
test also compares against the live reference on random layers.

Arithmetic model (identical to torch's CPU half/bfloat16 kernels): every
element-wise op converts its 16-bit operands to fp32, computes in fp32 and
rounds the result to the 16-bit type with round-to-nearest-even.  The final
contraction (third-party: torch ``F.linear``) is fp32-or-better accumulation
and ONE rounding to the 16-bit type; the oracle accumulates in float64.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Optional, Tuple

import numpy as np

__all__ = [
    "LayerSpec", "round_to", "to_f32", "from_f32",
    "unpack_indices", "pack_indices", "dequant", "gemv", "forward",
    "gemv_v2_ground_truth", "make_layer",
]


# --------------------------------------------------------------------------
# 16-bit float helpers.  Tensors are carried as uint16 BIT PATTERNS plus a
# dtype tag ("f16" | "bf16"); arithmetic happens in fp32.
# --------------------------------------------------------------------------
def to_f32(bits: np.ndarray, dtype: str) -> np.ndarray:
    bits = np.ascontiguousarray(bits).view(np.uint16)
    if dtype == "f16":
        return bits.view(np.float16).astype(np.float32)
    if dtype == "bf16":
        return (bits.astype(np.uint32) << np.uint32(16)).view(np.float32)
    raise ValueError(dtype)


def from_f32(x: np.ndarray, dtype: str) -> np.ndarray:
    """fp32 -> 16-bit bit pattern, round-to-nearest-even."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    if dtype == "f16":
        with np.errstate(over="ignore"):
            return x.astype(np.float16).view(np.uint16)
    if dtype == "bf16":
        u = x.view(np.uint32)
        nan = np.isnan(x)
        lsb = (u >> np.uint32(16)) & np.uint32(1)
        r = ((u + np.uint32(0x7FFF) + lsb) >> np.uint32(16)).astype(np.uint16)
        r[nan] = np.uint16(0x7FC0)
        return r
    raise ValueError(dtype)


def round_to(x: np.ndarray, dtype: str) -> np.ndarray:
    """Round an fp32 array to the 16-bit grid and come back as fp32."""
    return to_f32(from_f32(x, dtype), dtype)


# --------------------------------------------------------------------------
# Packed index wire format (pack.py:26-67 / :105-139).
#   indices : int32 [C, N, ceil(G*T/32)],  T = index_bits + res_bits
#   element g of row (c, n) = bits [g*T, (g+1)*T) of the little-endian stream
#   value = (res_idx << index_bits) | idx
# --------------------------------------------------------------------------
def unpack_indices(packed: np.ndarray, index_bits: int, num_elements: int,
                   res_bits: int = 0, ref_residual_mask_quirk: bool = True
                   ) -> Tuple[np.ndarray, Optional[np.ndarray]]:
    """-> (idx int64 [C,N,G], res_idx int64 [C,N,G] | None).

    ``ref_residual_mask_quirk``: the reference masks the residual index with
    ``(1 << index_bits) - 1`` (pack.py:137) instead of ``res_bits``; identical
    whenever res_bits <= index_bits (every shipped configuration).  The
    reference's native kernel uses the res_bits mask (quant_gemv.cuh:120-121);
    pass False for that behaviour.
    """
    packed = np.ascontiguousarray(packed).view(np.uint32)
    C, N, W = packed.shape
    T = index_bits + res_bits
    G = num_elements
    assert W * 32 >= G * T, "row too short"
    # bit b of the stream = bit b%32 of word b//32
    bitpos = np.arange(G, dtype=np.int64) * T            # first bit of elem g
    words = packed.astype(np.uint64)
    # pad one word so a straddling read never runs off the row
    words = np.concatenate([words, np.zeros((C, N, 2), np.uint64)], axis=-1)
    wi = (bitpos >> 5)
    sh = (bitpos & 31).astype(np.uint64)
    lo = words[..., wi]
    hi = words[..., wi + 1]
    window = lo | (hi << np.uint64(32))                     # 64-bit window
    val = (window >> sh) & np.uint64((1 << T) - 1)
    idx = (val & np.uint64((1 << index_bits) - 1)).astype(np.int64)
    res = None
    if res_bits > 0:
        mbits = index_bits if ref_residual_mask_quirk else res_bits
        res = ((val >> np.uint64(index_bits)) &
               np.uint64((1 << mbits) - 1)).astype(np.int64)
    return idx, res


def pack_indices(idx: np.ndarray, index_bits: int,
                 res_idx: Optional[np.ndarray] = None, res_bits: int = 0
                 ) -> np.ndarray:
    """(idx [C,N,G] , res_idx) -> int32 [C,N,ceil(G*T/32)] (pack.py:26-67)."""
    idx = np.asarray(idx).astype(np.uint64)
    C, N, G = idx.shape
    T = index_bits + res_bits
    assert T <= 32
    merged = idx & np.uint64((1 << index_bits) - 1)
    if res_idx is not None:
        merged = merged | (np.asarray(res_idx).astype(np.uint64)
                           << np.uint64(index_bits))
    W = (G * T + 31) // 32
    out = np.zeros((C, N, W + 1), dtype=np.uint64)
    bitpos = np.arange(G, dtype=np.int64) * T
    wi = bitpos >> 5
    sh = (bitpos & 31).astype(np.uint64)
    lo = (merged << sh) & np.uint64(0xFFFFFFFF)
    hi = (merged << sh) >> np.uint64(32)
    # elements never overlap, so OR == ADD; ufunc.at handles repeated words
    flat = out.reshape(C * N, W + 1)
    rows = np.arange(C * N)[:, None]
    np.bitwise_or.at(flat, (rows, wi[None, :]), lo.reshape(C * N, G))
    np.bitwise_or.at(flat, (rows, wi[None, :] + 1), hi.reshape(C * N, G))
    return out[..., :W].astype(np.uint32).view(np.int32)


# --------------------------------------------------------------------------
# Layer description (mirrors VQuantLinear's parameters, vqlinear.py:97-240).
# All float tensors are uint16 bit patterns of `dtype`.
# --------------------------------------------------------------------------
@dataclass
class LayerSpec:
    in_features: int
    out_features: int
    vector_len: int
    num_centroids: int
    num_res_centroids: int           # <=0: no residual codebook
    num_codebooks: int               # group_num
    group_size: int
    outlier_size: int = 0
    outlier_vector_len: int = -1
    num_outlier_centroids: int = -1
    dtype: str = "f16"
    # tensors
    indices: np.ndarray = None               # int32 [C, N, W]
    centroids: np.ndarray = None             # u16 [C, k, v]
    res_centroids: Optional[np.ndarray] = None   # u16 [C, kr, v]
    outlier_indices: Optional[np.ndarray] = None  # u16 [1, M, S]
    outlier_centroids: Optional[np.ndarray] = None  # u16 [1, ko, ov]
    perm: Optional[np.ndarray] = None        # u16 [I]
    weight_scale: Optional[np.ndarray] = None  # u16 [I]
    weight_bias: Optional[np.ndarray] = None   # u16 [I]
    bias: Optional[np.ndarray] = None        # u16 [O]
    extra: dict = field(default_factory=dict)

    @property
    def index_bits(self) -> int:
        return int(math.ceil(math.log2(self.num_centroids)))

    @property
    def res_bits(self) -> int:
        return (int(math.ceil(math.log2(self.num_res_centroids)))
                if self.num_res_centroids > 0 else 0)

    @property
    def padding(self) -> int:
        return (-self.out_features) % self.vector_len

    @property
    def num_indices(self) -> int:
        return (self.out_features + self.padding) // self.vector_len

    @property
    def enable_outlier(self) -> bool:
        return self.outlier_vector_len > 1 and self.num_outlier_centroids > 0

    @property
    def outlier_padding(self) -> int:
        return ((-self.out_features) % self.outlier_vector_len
                if self.enable_outlier else 0)


def dequant(L: LayerSpec, ref_residual_mask_quirk: bool = True) -> np.ndarray:
    """Dense W as uint16 bit patterns [O, I]  (quant_gemm.py:43-158).

    W[n*v+t, S + cb*G + g] = rnd(cent[cb, idx, t] + res[cb, ridx, t])
    W[m*ov+t, g]           = outl[0, oidx[0,m,g], t]            (g < S)
    W = W[:, argsort(perm)] ; W = rnd(rnd(W * scale) + wbias)
    """
    dt = L.dtype
    C, G, v = L.num_codebooks, L.group_size, L.vector_len
    N = L.num_indices
    idx, ridx = unpack_indices(L.indices, L.index_bits, G, L.res_bits,
                               ref_residual_mask_quirk)
    cent = to_f32(L.centroids, dt).reshape(C, L.num_centroids, v)
    # [C, N, G, v] -> W[n*v+t, cb*G+g]
    sel = np.take_along_axis(cent[:, :, None, :],
                             idx.reshape(C, N * G, 1, 1), axis=1)
    sel = sel.reshape(C, N, G, v)
    q = sel.transpose(1, 3, 0, 2).reshape(N * v, C * G)
    if L.num_res_centroids > 0:
        rc = to_f32(L.res_centroids, dt).reshape(C, L.num_res_centroids, v)
        rs = np.take_along_axis(rc[:, :, None, :],
                                ridx.reshape(C, N * G, 1, 1), axis=1)
        rs = rs.reshape(C, N, G, v).transpose(1, 3, 0, 2).reshape(N * v, C * G)
        q = round_to(q + rs, dt)
    if L.padding > 0:
        q = q[:-L.padding]
    if L.enable_outlier:
        ov, S = L.outlier_vector_len, L.outlier_size
        oc = to_f32(L.outlier_centroids, dt).reshape(
            L.num_outlier_centroids, ov)
        oi = np.ascontiguousarray(L.outlier_indices).view(np.uint16)
        oi = oi.reshape(-1, S).astype(np.int64)              # [M, S]
        M = oi.shape[0]
        qo = oc[oi]                                          # [M, S, ov]
        qo = qo.transpose(0, 2, 1).reshape(M * ov, S)
        if L.outlier_padding > 0:
            qo = qo[:-L.outlier_padding]
        q = np.concatenate([qo, q], axis=1)
    if L.perm is not None:
        p = np.ascontiguousarray(L.perm).view(np.uint16).astype(np.int64)
        q = q[:, np.argsort(p, kind="stable")]
    if L.weight_scale is not None and L.weight_bias is not None:
        s = to_f32(L.weight_scale, dt)[None, :]
        b = to_f32(L.weight_bias, dt)[None, :]
        q = round_to(round_to(q * s, dt) + b, dt)
    return from_f32(q, dt)


def gemv(W_bits: np.ndarray, x_bits: np.ndarray, dtype: str,
         bias_bits: Optional[np.ndarray] = None) -> np.ndarray:
    """y = F.linear(x, W, bias) (quant_gemm.py:274): float64 accumulate, one
    rounding to `dtype`.  x [..., I] -> y [..., O] (uint16 bit patterns)."""
    W = to_f32(W_bits, dtype).astype(np.float64)
    x = to_f32(x_bits, dtype).astype(np.float64)
    y = x @ W.T
    if bias_bits is not None:
        y = y + to_f32(bias_bits, dtype).astype(np.float64)
    return from_f32(y.astype(np.float32), dtype)


def gemv_f64(W_bits, x_bits, dtype, bias_bits=None) -> np.ndarray:
    """Unrounded float64 result (for error-budget reporting)."""
    W = to_f32(W_bits, dtype).astype(np.float64)
    x = to_f32(x_bits, dtype).astype(np.float64)
    y = x @ W.T
    if bias_bits is not None:
        y = y + to_f32(bias_bits, dtype).astype(np.float64)
    return y


def forward(L: LayerSpec, x_bits: np.ndarray,
            ref_residual_mask_quirk: bool = True) -> np.ndarray:
    """VQuantLinear.forward on the torch fallback (vqlinear.py:351-397)."""
    return gemv(dequant(L, ref_residual_mask_quirk), x_bits, L.dtype, L.bias)


# --------------------------------------------------------------------------
# v2 wire format (tests/test_quant_gemv.py:49-109): unpacked indices laid out
# [N][I] flat; W[row=i % I, col=(i // I)*v : +v] = cent[ids[i]] (+ res);
# W = scale * W + sbias (one rounding each in the reference's torch ops);
# out = x @ W (+ bias).
# --------------------------------------------------------------------------
def dequant_v2(indices: np.ndarray, centroids: np.ndarray,
               res_indices: Optional[np.ndarray],
               res_centroids: Optional[np.ndarray],
               scale: Optional[np.ndarray], sbias: Optional[np.ndarray],
               in_features: int, out_features: int, vector_len: int,
               dtype: str) -> np.ndarray:
    """-> W^T as fp32 [I, O] on the `dtype` grid."""
    v = vector_len
    I, O = in_features, out_features
    N = O // v
    ids = np.asarray(indices).reshape(N, I).astype(np.int64)
    cent = to_f32(centroids, dtype).reshape(-1, v)
    w = cent[ids]                                   # [N, I, v]
    if res_indices is not None:
        rids = np.asarray(res_indices).reshape(N, I).astype(np.int64)
        rc = to_f32(res_centroids, dtype).reshape(-1, v)
        w = round_to(w + rc[rids], dtype)
    w = w.transpose(1, 0, 2).reshape(I, O)          # [I, O]
    if scale is not None:
        s = to_f32(scale, dtype).reshape(I, 1)
        w = round_to(s * w, dtype)
    if sbias is not None:
        b = to_f32(sbias, dtype).reshape(I, 1)
        w = round_to(w + b, dtype)
    return w


def gemv_v2_ground_truth(x_bits, bias_bits, indices, centroids, res_indices,
                         res_centroids, scale, sbias, vector_len,
                         out_features, dtype) -> np.ndarray:
    x = to_f32(x_bits, dtype)
    I = x.shape[-1]
    wt = dequant_v2(indices, centroids, res_indices, res_centroids, scale,
                    sbias, I, out_features, vector_len, dtype)
    # `out[i, j, :] = vec @ weights` stores a `dtype` tensor: one rounding per output
    # (tests/test_quant_gemv.py:101-104), and only then `out += bias` (:106-107), a second
    # `dtype` op with its own rounding - pinned by tests/golden/v2/f16_k8192_r512_t2_bias.npz.
    y = round_to((x.astype(np.float64) @ wt.astype(np.float64)).astype(np.float32), dtype)
    if bias_bits is not None:
        y = round_to(y + to_f32(bias_bits, dtype).reshape(1, -1), dtype)
    return from_f32(y, dtype)


# --------------------------------------------------------------------------
# Synthetic layer factory shared by tests / smoke / bench (seeded, numpy RNG).
# dist "ref-test": normal(0.02, 0.5) everywhere (tests/test_quant_gemv.py:
# 129-151); "llm": centroids N(0,.02), residual N(0,.005), scale 1+.1N,
# wbias .01N.
# --------------------------------------------------------------------------
def make_layer(in_features: int, out_features: int, *, vector_len: int = 8,
               num_centroids: int = 256, num_res_centroids: int = 256,
               num_codebooks: int = 1, outlier_size: int = 0,
               outlier_vector_len: int = -1, num_outlier_centroids: int = -1,
               enable_norm: bool = True, enable_perm: bool = False,
               bias: bool = False, dtype: str = "f16", dist: str = "ref-test",
               seed: int = 1234) -> LayerSpec:
    rng = np.random.default_rng(seed)
    I, O, v, C = in_features, out_features, vector_len, num_codebooks
    S = outlier_size
    assert (I - S) % C == 0
    G = (I - S) // C

    def nrm(shape, mean, std):
        return from_f32((rng.standard_normal(shape) * std + mean)
                        .astype(np.float32), dtype)

    if dist == "ref-test":
        p = dict(c=(0.02, 0.5), r=(0.02, 0.5), s=(0.02, 0.5), b=(0.02, 0.5))
    else:
        p = dict(c=(0.0, 0.02), r=(0.0, 0.005), s=(1.0, 0.1), b=(0.0, 0.01))
    L = LayerSpec(I, O, v, num_centroids, num_res_centroids, C, G, S,
                  outlier_vector_len, num_outlier_centroids, dtype)
    N = L.num_indices
    idx = rng.integers(0, num_centroids, size=(C, N, G))
    ridx = None
    if num_res_centroids > 0:
        ridx = rng.integers(0, num_res_centroids, size=(C, N, G))
        L.res_centroids = nrm((C, num_res_centroids, v), *p["r"])
    L.indices = pack_indices(idx, L.index_bits, ridx, L.res_bits)
    L.centroids = nrm((C, num_centroids, v), *p["c"])
    if L.enable_outlier:
        M = (O + L.outlier_padding) // outlier_vector_len
        L.outlier_indices = rng.integers(
            0, num_outlier_centroids, size=(1, M, S)).astype(np.uint16)
        L.outlier_centroids = nrm((1, num_outlier_centroids,
                                   outlier_vector_len), *p["c"])
    if enable_perm:
        L.perm = rng.permutation(I).astype(np.uint16)
    if enable_norm:
        L.weight_scale = nrm((I,), *p["s"])
        L.weight_bias = nrm((I,), *p["b"])
    if bias:
        L.bias = nrm((O,), 0.0, 0.5 if dist == "ref-test" else 0.02)
    return L
