"""ctypes binding of the C oracle (oracle/vptq_oracle.c).  TEST INFRASTRUCTURE:
import only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_build", "libvptq_oracle.so")

_vp = C.c_void_p


class Desc(C.Structure):  # VptqLayerDesc with HOST pointers (include/vptq_hip.h)
    _fields_ = [(n, C.c_int32) for n in (
        "in_features", "out_features", "vector_len", "num_codebooks", "group_size",
        "num_centroids", "num_res_centroids", "index_bits", "res_bits", "row_words",
        "num_indices", "outlier_size", "outlier_vector_len", "num_outlier_centroids",
        "num_outlier_indices", "dtype")] + [(n, _vp) for n in (
            "indices", "centroids", "res_centroids", "outlier_indices", "outlier_centroids",
            "perm", "inv_perm", "weight_scale", "weight_bias", "bias", "scale_permuted",
            "bias_permuted", "prefetch")] + [("prefetch_bytes", C.c_int64)]


_lib = None


def available():
    return os.path.exists(LIB)


def lib():
    global _lib
    if _lib is None:
        l = C.CDLL(LIB)
        l.vo_dequant.restype = C.c_int
        l.vo_dequant.argtypes = [C.POINTER(Desc), _vp, C.c_int]
        l.vo_linear.restype = C.c_int
        l.vo_linear.argtypes = [_vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int]
        l.vo_forward.restype = C.c_int
        l.vo_forward.argtypes = [C.POINTER(Desc), _vp, _vp, C.c_int, _vp, C.c_int]
        l.vo_num_threads.restype = C.c_int
        l.vo_set_num_threads.argtypes = [C.c_int]
        l.vo_f32_to_f16.restype = C.c_uint16
        l.vo_f32_to_f16.argtypes = [C.c_float]
        l.vo_f16_to_f32.restype = C.c_float
        l.vo_f16_to_f32.argtypes = [C.c_uint16]
        l.vo_f32_to_bf16.restype = C.c_uint16
        l.vo_f32_to_bf16.argtypes = [C.c_float]
        _lib = l
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data


def make_desc(L):
    """oracle LayerSpec (numpy) -> (Desc, keepalive list)."""
    c = lambda a, dt: None if a is None else np.ascontiguousarray(a).view(dt)  # noqa: E731
    keep = dict(
        indices=c(L.indices, np.int32), centroids=c(L.centroids, np.uint16),
        res_centroids=c(L.res_centroids, np.uint16) if L.num_res_centroids > 0 else None,
        outlier_indices=c(L.outlier_indices, np.uint16) if L.enable_outlier else None,
        outlier_centroids=c(L.outlier_centroids, np.uint16) if L.enable_outlier else None,
        perm=c(L.perm, np.uint16), weight_scale=c(L.weight_scale, np.uint16),
        weight_bias=c(L.weight_bias, np.uint16), bias=c(L.bias, np.uint16))
    d = Desc()
    d.in_features, d.out_features, d.vector_len = L.in_features, L.out_features, L.vector_len
    d.num_codebooks, d.group_size = L.num_codebooks, L.group_size
    d.num_centroids = L.num_centroids
    d.num_res_centroids = max(L.num_res_centroids, 0)
    d.index_bits, d.res_bits = L.index_bits, L.res_bits
    d.row_words = keep["indices"].shape[-1]
    d.num_indices = L.num_indices
    if L.enable_outlier:
        d.outlier_size, d.outlier_vector_len = L.outlier_size, L.outlier_vector_len
        d.num_outlier_centroids = L.num_outlier_centroids
        d.num_outlier_indices = (L.out_features + L.outlier_padding) // L.outlier_vector_len
    d.dtype = 0 if L.dtype == "f16" else 1
    for k, v in keep.items():
        setattr(d, k, _p(v))
    return d, keep


def dequant(L, quirk=True):
    d, keep = make_desc(L)
    W = np.empty((L.out_features, L.in_features), dtype=np.uint16)
    rc = lib().vo_dequant(d, W.ctypes.data, int(quirk))
    assert rc == 0
    return W


def forward(L, x_bits, quirk=True, scratch=None):
    d, keep = make_desc(L)
    x = np.ascontiguousarray(x_bits).view(np.uint16)
    tokens = x.size // L.in_features
    y = np.empty(x.shape[:-1] + (L.out_features,), dtype=np.uint16)
    if scratch is None:
        scratch = np.empty((L.out_features, L.in_features), dtype=np.uint16)
    rc = lib().vo_forward(d, x.ctypes.data, y.ctypes.data, tokens, scratch.ctypes.data, int(quirk))
    assert rc == 0
    return y
