"""Load tests/golden/*.npz into oracle LayerSpecs (shared by CPU and GPU tests)."""
import glob
import json
import os

import numpy as np

from oracle import vptq_oracle as vo
from _proc import proc_values

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_names():
    return sorted(os.path.splitext(os.path.basename(p))[0]
                  for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


def _proc_codebook(spec, dtype):
    n = spec["n"]
    v = proc_values(n, spec["seed"], spec["scale"]) + np.float32(spec["mean"])
    return vo.from_f32(v, dtype)


def load_golden(name):
    """-> (LayerSpec, x_bits [1,T,I], y_bits [1,T,O], cfg dict, W_head)"""
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    cfg = json.loads(bytes(z["config"]).decode())
    dt = cfg["dtype"]
    C, k, kr, v = (cfg["num_codebooks"], cfg["num_centroids"],
                   cfg["num_res_centroids"], cfg["vector_len"])

    def cb(key, count):
        if key in cfg["proc"]:
            sp = dict(cfg["proc"][key], n=C * count * v)
            return _proc_codebook(sp, dt).reshape(C, count, v)
        return z[key].reshape(C, count, v)

    L = vo.LayerSpec(
        cfg["in_features"], cfg["out_features"], v, k, kr, C, cfg["group_size"],
        cfg["outlier_size"], cfg["outlier_vector_len"], cfg["num_outlier_centroids"], dt)
    L.indices = z["indices"]
    L.centroids = cb("centroids", k)
    if kr > 0:
        L.res_centroids = cb("res_centroids", kr)
    if L.enable_outlier:
        L.outlier_indices = z["outlier_indices"]
        L.outlier_centroids = z["outlier_centroids"].reshape(
            1, cfg["num_outlier_centroids"], cfg["outlier_vector_len"])
    if cfg["enable_perm"]:
        L.perm = z["perm"]
    if cfg["enable_norm"]:
        L.weight_scale = z["weight_scale"]
        L.weight_bias = z["weight_bias"]
    if cfg["has_bias"]:
        L.bias = z["bias"]
    return L, z["x"], z["y"], cfg, z["W_head"]


def rel_err(y_bits, ref_bits, dtype):
    """max|Δ| / max|ref|  (the BASELINE.md §5 parity metric)."""
    a = vo.to_f32(y_bits, dtype).astype(np.float64)
    b = vo.to_f32(ref_bits, dtype).astype(np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-30))


def bit_identical_frac(y_bits, ref_bits):
    a = np.ascontiguousarray(y_bits).view(np.uint16).ravel()
    b = np.ascontiguousarray(ref_bits).view(np.uint16).ravel()
    return float(np.mean(a == b))


# ---- reference goldens at BASELINE sizes (tests/golden/gen_golden_big.py) ------------------
BIG_DIR = os.path.join(GOLDEN_DIR, "big")


def big_names():
    return sorted(os.path.splitext(os.path.basename(p))[0]
                  for p in glob.glob(os.path.join(BIG_DIR, "*.npz")))


FMT_DIR = os.path.join(GOLDEN_DIR, "fmt")   # tests/golden/gen_golden_fmt.py: the other formats


def fmt_names():
    return sorted(os.path.splitext(os.path.basename(p))[0]
                  for p in glob.glob(os.path.join(FMT_DIR, "*.npz")))


def load_fmt(name):
    return load_big(name, FMT_DIR)


def load_big(name, directory=None):
    """-> (LayerSpec, x_bits [1,T,I], y_bits [1,T,O] from the real reference, cfg, W_head).
    Every input is rebuilt procedurally (tests/golden/_proc.py:big_tensors); the fixture holds
    the reference's output and the sha256 of its dense W."""
    from _proc import big_tensors
    z = np.load(os.path.join(directory or BIG_DIR, name + ".npz"))
    cfg = json.loads(bytes(z["config"]).decode())
    I, O, dt = cfg["in_features"], cfg["out_features"], cfg["dtype"]
    v, k, kr = cfg["vector_len"], cfg["num_centroids"], cfg["num_res_centroids"]
    t = big_tensors(I, O, v, k, kr, cfg["perm"], cfg["bias"], cfg["tokens"], dt, cfg["dist"],
                    cfg["seed"])
    L = vo.LayerSpec(I, O, v, k, kr, 1, I, 0, -1, -1, dt)
    L.indices = t["indices"]
    L.centroids = t["centroids"].reshape(1, k, v)
    L.res_centroids = t["res_centroids"].reshape(1, kr, v)
    L.weight_scale, L.weight_bias = t["weight_scale"], t["weight_bias"]
    if cfg["perm"]:
        L.perm = t["perm"]
    if cfg["bias"]:
        L.bias = t["bias"]
    # y rows the fixture keeps (gen_golden_big.py:stored_rows): all up to 64 tokens, every 8th + the last beyond
    T = cfg["tokens"]
    cfg["y_rows"] = list(range(T)) if T <= 64 else sorted(set(range(0, T, 8 if T <= 1024 else 256)) | {T - 1})
    return L, t["x"].reshape(1, cfg["tokens"], I), z["y"], cfg, z["W_head"]


def stored_rows(out_bits, cfg):
    """the token rows of a full output [1, T, O] that a big fixture stores"""
    return np.ascontiguousarray(out_bits)[:, cfg["y_rows"], :]


# ---- v2 wire format: outputs of the reference test file's ground_truth ----------------------
V2_DIR = os.path.join(GOLDEN_DIR, "v2")


def v2_names():
    return sorted(os.path.splitext(os.path.basename(p))[0]
                  for p in glob.glob(os.path.join(V2_DIR, "*.npz")))


def load_v2(name):
    """-> dict: cfg, uint16 bit patterns x / centroids / res_centroids / scale_weights /
    scale_bias / bias (or None) / y (the reference's ground truth), and the index arrays
    (the reference's cyclic pattern, tests/test_quant_gemv.py:21-31)."""
    z = np.load(os.path.join(V2_DIR, name + ".npz"))
    cfg = json.loads(bytes(z["config"]).decode())
    n = cfg["in_features"] * cfg["out_features"] // cfg["vector_len"]
    k, kr = cfg["num_centroids"], cfg["num_res_centroids"]
    d = {key: z[key] for key in ("x", "centroids", "res_centroids", "scale_weights", "scale_bias", "y")}
    d["bias"] = z["bias"] if "bias" in z.files else None
    d["cfg"] = cfg
    d["indices"] = np.tile(np.arange(k, dtype=np.int64), n // k).astype(np.uint16)
    d["res_indices"] = np.tile(np.arange(kr, dtype=np.int64), n // kr).astype(
        np.uint16 if cfg["res_index_dtype"] == "uint16" else np.uint8)
    return d
