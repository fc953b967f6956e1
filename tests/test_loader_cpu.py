"""vptq.AutoModelForCausalLM.from_pretrained on a synthetic checkpoint (CPU: structure + tensors)."""
import pytest
import torch

import vptq_amd
from _ckpt import write_tiny_checkpoint


def test_loader_builds_and_loads_synthetic_checkpoint(tmp_path):
    state, per_layer = write_tiny_checkpoint(str(tmp_path))
    model = vptq_amd.AutoModelForCausalLM.from_pretrained(str(tmp_path), device="cpu")
    qmods = {n: m for n, m in model.named_modules() if isinstance(m, vptq_amd.VQuantLinear)}
    assert sorted(qmods) == sorted(per_layer) and len(qmods) == 2 * 7
    sd = model.state_dict()
    for k, v in state.items():
        assert k in sd and sd[k].dtype == v.dtype and torch.equal(sd[k].cpu(), v), k
    assert not any(p.is_meta for p in model.parameters())
    # rotary buffer rebuilt (not meta / garbage)
    inv = [b for n, b in model.named_buffers() if n.endswith("inv_freq")][0]
    assert torch.isfinite(inv).all() and inv[0] == 1.0
    assert model.lm_head.weight.dtype == torch.float16 and isinstance(model.lm_head, torch.nn.Linear)
    assert not model.training


def test_loader_absorb_perm_option(tmp_path):
    write_tiny_checkpoint(str(tmp_path))
    a = vptq_amd.AutoModelForCausalLM.from_pretrained(str(tmp_path), device="cpu")
    b = vptq_amd.AutoModelForCausalLM.from_pretrained(str(tmp_path), device="cpu", absorb_perm=True)
    qa = [m for m in a.modules() if isinstance(m, vptq_amd.VQuantLinear)]
    qb = [m for m in b.modules() if isinstance(m, vptq_amd.VQuantLinear)]
    assert all(m.enable_perm for m in qa) and not any(m.enable_perm for m in qb)
    assert all("perm" not in m.state_dict() for m in qb)
    cfg = b.config.quantization_config["config_for_layers"]
    assert not any(v["enable_perm"] for v in cfg.values())
    # same dense weight (oracle) for one layer
    from oracle import vptq_oracle as vo
    import numpy as np
    def spec(m):
        L = vo.LayerSpec(m.in_features, m.out_features, 8, 256, 256, 1, m.group_size, dtype="f16")
        u16 = lambda t: t.detach().contiguous().view(torch.int16).numpy().view(np.uint16)
        L.indices = m.indices.detach().numpy()
        L.centroids = u16(m.centroids.weight).reshape(1, 256, 8)
        L.res_centroids = u16(m.res_centroids.weight).reshape(1, 256, 8)
        L.weight_scale, L.weight_bias = u16(m.weight_scale), u16(m.weight_bias)
        L.perm = u16(m.perm) if m.enable_perm else None
        return L
    assert (vo.dequant(spec(qa[3])) == vo.dequant(spec(qb[3]))).all()


def test_loader_rejects_non_vptq_and_missing(tmp_path):
    import json, pytest
    with pytest.raises(FileNotFoundError):
        vptq_amd.AutoModelForCausalLM.from_pretrained(str(tmp_path / "nope"))
    write_tiny_checkpoint(str(tmp_path))
    cfg = json.load(open(tmp_path / "config.json"))
    del cfg["quantization_config"]
    json.dump(cfg, open(tmp_path / "config.json", "w"))
    with pytest.raises(ValueError, match="quantization_config"):
        vptq_amd.AutoModelForCausalLM.from_pretrained(str(tmp_path), device="cpu")


def test_loader_device_map_is_honoured_or_refused(tmp_path):
    """accelerate's multi-device placement (reference vptq/layers/model_base.py:165-194) is not carried over:
    a device_map that names one device is honoured, one that spreads the model is refused, nothing is silently
    swallowed"""
    import warnings
    write_tiny_checkpoint(str(tmp_path))
    m = vptq_amd.AutoModelForCausalLM.from_pretrained(str(tmp_path), device_map="cpu")
    assert next(m.parameters()).device.type == "cpu"
    m = vptq_amd.AutoModelForCausalLM.from_pretrained(str(tmp_path), device_map={"": "cpu"})
    assert next(m.parameters()).device.type == "cpu"
    with pytest.raises(NotImplementedError):
        vptq_amd.AutoModelForCausalLM.from_pretrained(str(tmp_path), device_map={"model.layers.0": "cuda:0", "lm_head": "cuda:1"})
    # HF's index forms: {"": 0} and a bare 1 are "cuda:0" / "cuda:1", never the invalid device string "0"
    from vptq_amd.layers.model_base import _device_from_map
    assert _device_from_map({"": 0}, None) == "cuda:0"
    assert _device_from_map(1, None) == "cuda:1"
    assert _device_from_map("1", None) == "cuda:1"
    assert _device_from_map({"": torch.device("cuda", 2)}, None) == "cuda:2"
    assert _device_from_map({"a": 0, "b": "cuda:0"}, None) == "cuda:0"
    assert _device_from_map("auto", None) is None and _device_from_map(None, "cpu") == "cpu"
    assert _device_from_map({"": 0}, "cpu") == "cpu"
    with pytest.raises(NotImplementedError):
        _device_from_map({"a": 0, "b": 1}, None)
    with pytest.raises(TypeError):
        _device_from_map(1.5, None)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        vptq_amd.AutoModelForCausalLM.from_pretrained(str(tmp_path), device="cpu", low_cpu_mem_usage=True)
    assert any("ignores" in str(x.message) for x in w)
