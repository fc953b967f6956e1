"""Offline packing of an unpacked checkpoint: `vptq_amd.utils.pack.pack_state_dict` against the
reference's `pack_model` (vptq/utils/pack.py:142-283) run on a stand-in model, and against the
oracle's bit-stream packer."""
import sys

import numpy as np
import pytest
import torch

from oracle import vptq_oracle as vo
from vptq_amd.utils.pack import pack_layer_tensors, pack_state_dict, unpack_index_tensor
from _refshim import reference_available

LAYERS = {
    # name: (C, N, G, k, k_res, as_float, perm, outliers)
    "model.layers.0.self_attn.q_proj": (1, 5, 64, 256, 256, True, True, False),
    "model.layers.0.mlp.up_proj": (2, 3, 40, 4096, 0, False, False, True),
    "model.layers.1.mlp.down_proj": (1, 4, 24, 65536, 16, True, False, False),
}


def unpacked_layer(seed, C, N, G, k, kr, as_float, perm, outliers):
    """what the quantisation algorithm stores: uint16 bit patterns in a float16 / int16 tensor,
    or plain int64 values"""
    rng = np.random.default_rng(seed)
    idx = rng.integers(0, k, size=(C, N, G))
    res = rng.integers(0, kr, size=(C, N, G)) if kr else None

    def store(a, as_f):
        t = torch.from_numpy(a.astype(np.uint16).view(np.int16))
        return t.view(torch.float16) if as_f else torch.from_numpy(a.astype(np.int64))
    d = {"indices": store(idx, as_float), "res_indices": None if res is None else store(res, as_float),
         "outlier_indices": None, "perm": None}
    if perm:
        d["perm"] = torch.from_numpy(rng.permutation(C * G).astype(np.int64))
    if outliers:
        d["outlier_indices"] = store(rng.integers(0, 64, size=(1, N * 2, 4)), True)
    return d, idx, res


def test_pack_layer_matches_the_oracle_bit_stream():
    for i, (name, (C, N, G, k, kr, as_f, perm, outl)) in enumerate(LAYERS.items()):
        d, idx, res = unpacked_layer(i, C, N, G, k, kr, as_f, perm, outl)
        out = pack_layer_tensors(d, k, kr if kr else 1)
        ib, rb = int(np.log2(k)), int(np.log2(kr)) if kr else 0
        assert (out["indices"].numpy() == vo.pack_indices(idx, ib, res, rb)).all()
        assert "res_indices" not in out and out["indices"].dtype == torch.int32
        a, b = unpack_index_tensor(out["indices"], ib, G, rb, G)
        assert (a.numpy() == idx).all() and (res is None or rb > ib or (b.numpy() == res).all())
        if perm:
            assert out["perm"].dtype == torch.int16
            assert (out["perm"].view(torch.int16).to(torch.int64) & 0xFFFF).tolist() == d["perm"].tolist()


def _state_and_config():
    state, conf = {}, {}
    for i, (name, (C, N, G, k, kr, as_f, perm, outl)) in enumerate(LAYERS.items()):
        d, _, _ = unpacked_layer(i, C, N, G, k, kr, as_f, perm, outl)
        for key, t in d.items():
            if t is not None:
                state[f"{name}.{key}"] = t
        state[f"{name}.centroids.weight"] = torch.zeros(C, 8)
        conf[name] = dict(in_features=C * G, out_features=N * 8, vector_lens=[-1, 8],
                          num_centroids=[-1, k], num_res_centroids=[-1, kr if kr else -1],
                          group_num=C, group_size=G, outlier_size=0, indices_as_float=as_f,
                          enable_norm=True, enable_perm=perm, is_indice_packed=False, bias=False)
    state["lm_head.weight"] = torch.ones(3, 3)
    return state, conf


def test_pack_state_dict_whole_checkpoint():
    state, conf = _state_and_config()
    new_state, new_conf = pack_state_dict(state, conf)
    assert torch.equal(new_state["lm_head.weight"], state["lm_head.weight"])
    for name in LAYERS:
        assert new_conf[name]["is_indice_packed"] is True and conf[name]["is_indice_packed"] is False
        assert f"{name}.res_indices" not in new_state
        C, N, G, k, kr = LAYERS[name][:5]
        T = int(np.log2(k)) + (int(np.log2(kr)) if kr else 0)
        assert tuple(new_state[f"{name}.indices"].shape) == (C, N, (G * T + 31) // 32)
    with pytest.raises(KeyError):
        pack_state_dict({}, conf)


@pytest.mark.skipif(not reference_available(), reason="/root/reference not mounted (GPU box)")
def test_pack_state_dict_equals_reference_pack_model():
    from _refshim import load_reference
    saved = {k: v for k, v in sys.modules.items() if k == "vptq" or k.startswith("vptq.")}
    try:
        for k in saved:
            sys.modules.pop(k)
        load_reference()
        from vptq.utils.pack import convert_idx_dtype   # pack_model = this + a HF config fix-up

        class VQuantLinear(torch.nn.Module):       # the reference finds layers by type name
            pass

        class Model(torch.nn.Module):
            class config:                          # noqa: N801 - stands in for a HF config
                pass

        state, conf = _state_and_config()
        model = Model()
        mods = {}
        for name, (C, N, G, k, kr, as_f, perm, outl) in LAYERS.items():
            m = VQuantLinear()
            for key in ("indices", "res_indices", "outlier_indices", "perm"):
                t = state.get(f"{name}.{key}")
                setattr(m, key, None if t is None else torch.nn.Parameter(t.clone(), requires_grad=False))
            if m.perm is None:                      # the reference converts perm unconditionally
                m.perm = torch.nn.Parameter(torch.arange(C * G), requires_grad=False)
            m.num_centroids, m.num_res_centroids = k, (kr if kr else -1)
            m.init_args = dict(conf[name])
            mods[name] = m
        # nest the modules under their dotted names
        for name, m in mods.items():
            parent = model
            parts = name.split(".")
            for p in parts[:-1]:
                if not hasattr(parent, p):
                    parent.add_module(p, torch.nn.Module())
                parent = getattr(parent, p)
            parent.add_module(parts[-1], m)
        packed = convert_idx_dtype(model, torch.uint16, torch.uint16, torch.int16)
        ref_conf = Model.config.quantization_config["config_for_layers"]
    finally:
        for k in [k for k in sys.modules if k == "vptq" or k.startswith("vptq.")]:
            sys.modules.pop(k)
        sys.modules.update(saved)
    new_state, new_conf = pack_state_dict(state, conf)
    for name, m in mods.items():
        assert torch.equal(m.indices.data, new_state[f"{name}.indices"]), name
        assert m.res_indices is None
        if state.get(f"{name}.perm") is not None:
            assert torch.equal(m.perm.data, new_state[f"{name}.perm"])
        if state.get(f"{name}.outlier_indices") is not None:
            assert torch.equal(m.outlier_indices.data, new_state[f"{name}.outlier_indices"])
        assert ref_conf[name]["is_indice_packed"] is True and new_conf[name] == ref_conf[name]
    assert packed is model
