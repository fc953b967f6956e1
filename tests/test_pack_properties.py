"""Property tests (hypothesis) of the packed-index wire format: host tools vs oracle."""
import numpy as np
import torch
from hypothesis import given, settings, strategies as st

from oracle import vptq_oracle as vo
from vptq_amd.utils.pack import pack_index, unpack_index_tensor


@settings(max_examples=60, deadline=None)
@given(ib=st.integers(1, 16), rb=st.integers(0, 16), G=st.integers(1, 70), C=st.integers(1, 3),
       N=st.integers(1, 4), seed=st.integers(0, 2**31 - 1))
def test_pack_unpack_roundtrip_any_width(ib, rb, G, C, N, seed):
    rng = np.random.default_rng(seed)
    idx = rng.integers(0, 1 << ib, size=(C, N, G))
    r = rng.integers(0, 1 << rb, size=(C, N, G)) if rb else None
    want = vo.pack_indices(idx, ib, r, rb)
    ti = torch.from_numpy(idx.astype(np.uint16).view(np.int16))
    tr = torch.from_numpy(r.astype(np.uint16).view(np.int16)) if rb else None
    got = pack_index(ti, ib, tr, rb)
    assert got.shape == (C, N, (G * (ib + rb) + 31) // 32)
    assert (got.numpy() == want).all()
    # stream layout: element g sits at bits [g*T, (g+1)*T) of the row, little endian
    T = ib + rb
    row = got.numpy().view(np.uint32)[0, 0]
    big = sum(int(w) << (32 * i) for i, w in enumerate(row))
    for g in range(G):
        v = (big >> (g * T)) & ((1 << T) - 1)
        assert v & ((1 << ib) - 1) == idx[0, 0, g]
        if rb:
            assert v >> ib == r[0, 0, g]
    a, b = unpack_index_tensor(got, ib, G, rb, G)
    assert (a.numpy() == idx).all()
    if rb and rb <= ib:            # reference quirk: residual masked with index_bits
        assert (b.numpy() == r).all()
    ao, bo = vo.unpack_indices(want, ib, G, rb, ref_residual_mask_quirk=False)
    assert (ao == idx).all() and (rb == 0 or (bo == r).all())
