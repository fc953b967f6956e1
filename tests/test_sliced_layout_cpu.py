"""The load-time derived layout of the k = 65536 formats (vptq_amd/utils/sliced.py:build_sliced_layout, consumed by
gemv_sliced.hip) on CPU tensors: every element of every row appears exactly once, in the slice of its index, with its
column, its index inside the slice and its residual index; lists are padded to blocks of 64 with (column = G, local 0);
`blocks` / `first` describe the contiguous per-(slice, row) streams the kernel walks.
Packed index format: vptq/utils/pack.py:26-89 (little-endian bit stream per row, element g at bits [T g, T g + T))."""
import numpy as np
import pytest
import torch

from vptq_amd.utils.sliced import build_sliced_layout, rows_per_wave_for


def _pack(idx, ridx):
    """idx [N, G] uint16 (+ ridx [N, G] uint8 or None) -> int32 [1, N, row_words]"""
    N, G = idx.shape
    nb = 2 if ridx is None else 3
    by = np.zeros((N, G, nb), np.uint8)
    by[:, :, 0] = idx & 0xff
    by[:, :, 1] = idx >> 8
    if ridx is not None:
        by[:, :, 2] = ridx
    flat = by.reshape(N, G * nb)
    pad = (-flat.shape[1]) % 4
    if pad:
        flat = np.concatenate([flat, np.zeros((N, pad), np.uint8)], axis=1)
    return torch.from_numpy(np.ascontiguousarray(flat).view(np.int32).reshape(1, N, -1).copy())


@pytest.mark.parametrize("N,G,slices,residual,dist", [
    (8, 64, 8, False, "uniform"),
    (21, 520, 8, True, "uniform"),
    (16, 2048, 16, True, "uniform"),
    (5, 1000, 8, False, "one-slice"),      # every index in slice 3: the other slices' lists are empty
    (7, 136, 16, True, "one-entry"),       # every element the same index: one class, one slice
    (33, 72, 8, True, "uniform"),
])
def test_sliced_layout_holds_every_element_once(N, G, slices, residual, dist):
    rng = np.random.default_rng(N * 1000 + G)
    if dist == "uniform":
        idx = rng.integers(0, 65536, (N, G), dtype=np.int64)
    elif dist == "one-slice":
        idx = 3 * (65536 // slices) + rng.integers(0, 65536 // slices, (N, G), dtype=np.int64)
    else:
        idx = np.full((N, G), 40000, np.int64)
    idx = idx.astype(np.uint16)
    ridx = rng.integers(0, 256, (N, G)).astype(np.uint8) if residual else None
    elems, blocks, first, res, wstart = build_sliced_layout(_pack(idx, ridx), G, slices, residual)
    from vptq_amd.utils.sliced import WINDOWS, window_cols
    assert wstart.dtype == torch.int32 and wstart.shape == (slices, N, WINDOWS + 1)
    wstart, wc = wstart.numpy(), window_cols(G)
    assert wc % 8 == 0 and wc * WINDOWS >= G
    assert elems.dtype == torch.int32 and blocks.shape == (slices, N) and first.shape == (slices, N)
    assert (res is None) == (not residual)
    e = elems.numpy().view(np.uint32)
    blocks, first = blocks.numpy(), first.numpy()
    slice_size = 65536 // slices
    # `first` is the running total of `blocks` in (slice, row) order: one contiguous stream per workgroup's wave
    flat_b = blocks.reshape(-1)
    assert np.array_equal(first.reshape(-1), np.cumsum(flat_b) - flat_b)
    assert e.size == max(int(flat_b.sum()), 1) * 64
    if residual:
        assert res.dtype == torch.uint8 and res.numel() == e.size
        r = res.numpy()
    for n in range(N):
        seen = np.zeros(G, bool)
        for s in range(slices):
            lo, cnt = int(first[s, n]) * 64, int(blocks[s, n]) * 64
            w = e[lo:lo + cnt]
            col, local = w & 0xffff, w >> 16
            real = col < G
            # padding: (column G, local 0) only, and only at the end of the list, less than one block of it
            assert np.all(col[~real] == G) and np.all(local[~real] == 0)
            k = int(real.sum())
            assert np.all(real[:k]) and cnt - k < 64 and (cnt == 0) == (k == 0)
            c = col[:k].astype(np.int64)
            assert not seen[c].any() and len(np.unique(c)) == k       # each column once
            seen[c] = True
            assert np.all(local[:k] < slice_size)
            assert np.array_equal(s * slice_size + local[:k].astype(np.int64), idx[n, c].astype(np.int64))
            if residual:
                assert np.array_equal(r[lo:lo + k], ridx[n, c])
            # ordered by column window; wstart[s, n, w] = where window w begins, [WINDOWS] = the list's length
            ws = wstart[s, n]
            assert ws[0] == 0 and ws[WINDOWS] == k and np.all(np.diff(ws) >= 0)
            assert np.array_equal(np.minimum(c // wc, WINDOWS - 1), np.repeat(np.arange(WINDOWS), np.diff(ws)))
        assert seen.all()


def test_sliced_layout_spreads_lds_bank_groups():
    """inside a (row, slice, column window) list, 16 consecutive elements - one pass of the kernel's ds_read_b128 gather -
    should hit different bank groups (entry & 15) as far as the list's classes allow: while every class still has an
    element left (rank-major order: the first min-class-count rows of 16) a group of 16 holds 16 different classes, and
    overall the mean multiplicity stays well under column order's (3-way conflicts on average)"""
    from vptq_amd.utils.sliced import WINDOWS
    rng = np.random.default_rng(7)
    N, G = 4, 8192
    idx = rng.integers(0, 65536, (N, G)).astype(np.uint16)
    elems, blocks, first, _, wstart = build_sliced_layout(_pack(idx, None), G, 8, False)
    e = elems.numpy().view(np.uint32)
    wstart = wstart.numpy()
    mult, groups, free = 0, 0, 0
    for n in range(N):
        for s in range(8):
            lo = int(first[s, n]) * 64
            for w in range(WINDOWS):
                a, b = int(wstart[s, n, w]), int(wstart[s, n, w + 1])
                cls = (e[lo + a:lo + b] >> 16) & 15
                full = int(np.bincount(cls, minlength=16).min())      # rows of 16 that hold every class
                assert full >= 4
                for r in range(full):
                    assert len(np.unique(cls[16 * r:16 * r + 16])) == 16
                free += full * 16
            k = int(wstart[s, n, WINDOWS])
            cls = (e[lo:lo + k] >> 16) & 15
            for i in range(0, k - 15, 16):   # (the kernel's groups of 16 lanes are aligned to the list, not to its windows)
                mult += int(np.bincount(cls[i:i + 16], minlength=16).max())
                groups += 1
    assert free >= 0.5 * N * G and mult / groups <= 1.8, (free / (N * G), mult / groups)


def test_rows_per_wave():
    assert rows_per_wave_for(1024, 8) == 2       # 8192 rows of 8 outputs: 8 slices x 32 row blocks = 256 workgroups
    assert rows_per_wave_for(512, 8) == 1
    assert rows_per_wave_for(3584, 16) == 14
    assert rows_per_wave_for(10 ** 6, 8) == 64   # capped: block counts of a wave's rows sit in the lanes of one register


def test_two_table_format_splits_into_two_index_streams():
    """v8-k65536-65536 (T = 32): element g = (residual index << 16) | main index, one 32-bit word; the sliced path runs
    one pass per table, each over a layout built from THAT table's index stream"""
    from vptq_amd.utils.sliced import split_index_streams, layout_from_indices
    rng = np.random.default_rng(5)
    N, G = 9, 200
    idx = rng.integers(0, 65536, (N, G), dtype=np.int64)
    ridx = rng.integers(0, 65536, (N, G), dtype=np.int64)
    words = ((ridx << 16) | idx).astype(np.uint32).view(np.int32).reshape(1, N, G)
    a, b = split_index_streams(torch.from_numpy(words.copy()), G, 16)
    assert np.array_equal(a.numpy(), idx) and np.array_equal(b.numpy(), ridx)
    a8, b8 = split_index_streams(_pack(idx.astype(np.uint16), (ridx & 255).astype(np.uint8)), G, 8)
    assert np.array_equal(a8.numpy(), idx) and np.array_equal(b8.numpy(), ridx & 255)
    # the residual stream's layout holds every element once, in the slice of its RESIDUAL index
    elems, blocks, first, res, _ = layout_from_indices(b, 8)
    assert res is None
    e = elems.numpy().view(np.uint32)
    for n in range(N):
        got = {}
        for s in range(8):
            lo, cnt = int(first[s, n]) * 64, int(blocks[s, n]) * 64
            w = e[lo:lo + cnt]
            for col, local in zip(w & 0xffff, w >> 16):
                if col < G:
                    got[int(col)] = s * 8192 + int(local)
        assert got == {g: int(ridx[n, g]) for g in range(G)}


@pytest.mark.parametrize("V", [8, 16])
def test_matrix_pipe_operand_mapping_emulated(V):
    """gemv_sliced_tok.hip, 3 - 4 tokens: the contraction over a block's 64 elements on the matrix pipe.  numpy emulation of
    what the kernel relies on - ds_read_b64_tr_b16 (inside 16 lanes, source lane 4 e + c supplies an 8-byte chunk, result lane
    4 c' + m receives half m of the chunks of source lanes 4 e' + c', e' = 0..3) and the operand layout of v_mfma_f32_16x16x32
    (lane 16 g + j: row / column j, K = 8 g .. 8 g + 7; D: lane 16 g + j holds rows 4 g .. 4 g + 3 of column j) - with the
    kernel's choice of element and chunk per source lane: the sums land where row_end() reads them."""
    rng = np.random.default_rng(V)
    E = rng.standard_normal((64, V))           # the block's entries
    X = rng.standard_normal((64, 4))           # its activations, 4 tokens
    reads = 2 if V == 8 else 4
    per_read = 64 // reads                     # elements a read covers
    chunks = V // 4                            # 8-byte chunks of an entry

    def source(lane, i):                       # (element position in the block, chunk) of source lane `lane` in read i
        pos = per_read * i + (lane >> 1 if V == 8 else lane >> 2)
        return pos, lane & (chunks - 1)

    def tr_read(chunk_of_lane):                # chunk_of_lane[lane] = 4 values -> result[lane] = 4 values
        out = np.zeros((64, 4))
        for g in range(4):
            for c in range(4):
                for m in range(4):
                    out[16 * g + 4 * c + m] = [chunk_of_lane[16 * g + 4 * e + c][m] for e in range(4)]
        return out
    A = np.zeros((16, 32 * (reads // 2)))      # rows: 8 h + token (v = 16: token), K = (MFMA, lane group, read, e')
    Bm = np.zeros((32 * (reads // 2), 16))     # columns: 8 h + component (v = 16: component)
    for i in range(reads):
        ent, act = [], []
        for lane in range(64):
            pos, q = source(lane, i)
            ent.append(E[pos, 4 * q:4 * q + 4])
            act.append(X[pos] if q == 0 else np.zeros(4))       # chunk 0's lanes bring the tokens, the others the zero column
        bt, at = tr_read(ent), tr_read(act)
        for lane in range(64):
            g, j = lane >> 4, lane & 15
            for e in range(4):
                k = 32 * (i // 2) + 8 * g + 4 * (i % 2) + e     # reads 2 m, 2 m + 1 feed MFMA m
                A[j, k] = at[lane][e]
                Bm[k, j] = bt[lane][e]
    D = A @ Bm                                  # (v = 16: the two MFMAs accumulate into the same D)
    want = E.T @ X                              # [component][token]
    if V == 8:    # lanes 0 - 7 hold set 0 (rows 0 - 3 = tokens), lanes 40 - 47 set 1 (rows 8 - 11, columns 8 - 15)
        got = np.array([[D[t, comp] + D[8 + t, 8 + comp] for t in range(4)] for comp in range(8)])
    else:         # lanes 0 - 15: rows 0 - 3 = tokens, column = component
        got = np.array([[D[t, comp] for t in range(4)] for comp in range(16)])
    assert np.allclose(got, want)
    # every element is taken exactly once per chunk
    seen = sorted(source(lane, i) for i in range(reads) for lane in range(64))
    assert seen == sorted((p, q) for p in range(64) for q in range(chunks))
