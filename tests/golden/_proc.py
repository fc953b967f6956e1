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
