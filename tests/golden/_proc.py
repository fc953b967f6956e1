"""Procedural (LCG) codebooks so 1-MiB k=65536 tables need not be stored.

Used by gen_golden.py (to fill the reference module) and by the tests (to
rebuild the identical bit patterns).  Pure integer arithmetic -> reproducible
across numpy versions.
"""
import numpy as np

PROC_MIN_ELEMS = 32768  # codebooks with >= this many scalars are procedural


def lcg_u32(n: int, seed: int) -> np.ndarray:
    """n values of a 64-bit LCG (Knuth MMIX constants), top 32 bits."""
    a = np.uint64(6364136223846793005)
    c = np.uint64(1442695040888963407)
    out = np.empty(n, dtype=np.uint32)
    # jump-free vectorised form: process in blocks using cumulative affine maps
    s = np.uint64(seed * 2 + 1)
    with np.errstate(over="ignore"):
        for i in range(n):
            s = s * a + c
            out[i] = np.uint32(s >> np.uint64(32))
    return out


def proc_values(n: int, seed: int, scale: float) -> np.ndarray:
    """fp32 values in [-scale, scale) on a 2^-11 grid (exact in fp16/bf16? no:
    callers round to the layer dtype themselves)."""
    u = lcg_u32_fast(n, seed)
    return (((u >> np.uint32(20)).astype(np.int64) - 2048) / 2048.0 * scale).astype(np.float32)


def lcg_u32_fast(n: int, seed: int) -> np.ndarray:
    """Same stream as lcg_u32, vectorised by squaring the affine map."""
    a = np.uint64(6364136223846793005)
    c = np.uint64(1442695040888963407)
    with np.errstate(over="ignore"):
        # state_i = A_i * s0 + C_i ; build A_i, C_i for i=1..n by doubling
        A = np.empty(n, dtype=np.uint64)
        C = np.empty(n, dtype=np.uint64)
        A[0], C[0] = a, c
        filled = 1
        while filled < n:
            m = min(filled, n - filled)
            # map for (i + filled) = map_filled ∘ map_i  => A = A_f*A_i, C = A_f*C_i + C_f
            Af, Cf = A[filled - 1], C[filled - 1]
            A[filled:filled + m] = Af * A[:m]
            C[filled:filled + m] = Af * C[:m] + Cf
            filled += m
        s0 = np.uint64(seed * 2 + 1)
        s = A * s0 + C
    return (s >> np.uint64(32)).astype(np.uint32)


def _dtype_bits(v: np.ndarray, dtype: str) -> np.ndarray:
    """fp32 -> uint16 bit patterns of fp16 / bf16 (round to nearest even).  Every value this
    module produces lies on a 2^-11 grid times a short scale, far from overflow."""
    v = np.ascontiguousarray(v, dtype=np.float32)
    if dtype == "f16":
        return v.astype(np.float16).view(np.uint16)
    u = v.view(np.uint32)
    return ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) >> np.uint32(16)).astype(np.uint16)


def big_tensors(I, O, v, k, kr, perm, bias, tokens, dtype, dist, seed):
    """Every input of a canonical-format layer of BASELINE size, procedurally (integer LCG):
    the same bits in gen_golden_big.py (which feeds them to the real reference) and in the
    tests.  Returns uint16 bit patterns (int32 for the index words)."""
    N = (O + v - 1) // v
    T = int(np.log2(k)) + (int(np.log2(kr)) if kr > 0 else 0)   # (kr <= 0: no residual codebook)
    W = (I * T + 31) // 32
    p = (dict(c=(0.02, 0.5), r=(0.02, 0.5), s=(0.02, 0.5), b=(0.02, 0.5), x=(0.02, 0.5), o=(0.02, 0.5))
         if dist == "ref-test" else
         dict(c=(0.0, 0.02), r=(0.0, 0.005), s=(1.0, 0.1), b=(0.0, 0.01), x=(0.0, 1.0), o=(0.0, 0.02)))

    def vals(n, sd, key):
        mean, std = p[key]
        return _dtype_bits(proc_values(n, sd, 3 * std) + np.float32(mean), dtype)

    out = dict(
        indices=lcg_u32_fast(N * W, seed * 16 + 1).view(np.int32).reshape(1, N, W),
        centroids=vals(k * v, seed * 16 + 2, "c"),
        res_centroids=vals(kr * v, seed * 16 + 3, "r") if kr > 0 else np.zeros(0, dtype=np.uint16),
        weight_scale=vals(I, seed * 16 + 4, "s"), weight_bias=vals(I, seed * 16 + 5, "b"),
        x=vals(tokens * I, seed * 16 + 6, "x"))
    if perm:
        # a permutation from the LCG: argsort of distinct 64-bit keys (value << 16 | position)
        key = (lcg_u32_fast(I, seed * 16 + 7).astype(np.uint64) << np.uint64(16)) | np.arange(I, dtype=np.uint64)
        out["perm"] = np.argsort(key, kind="stable").astype(np.uint16)
    if bias:
        out["bias"] = vals(O, seed * 16 + 8, "o")
    return out
