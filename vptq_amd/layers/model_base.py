"""Checkpoint loader: `vptq.AutoModelForCausalLM.from_pretrained` for VPTQ checkpoints
(reference vptq/layers/model_base.py:93-199).

Same observable behaviour for the single-device case: read `config.json`, build the model
without allocating weights, swap every module named in `quantization_config.config_for_layers`
(or whose tail name is in `shared_layer_config`, model_base.py:41-45) for a `VQuantLinear`
built from that entry's keyword arguments, load the safetensors shards, return `model.eval()`.

Not carried over: accelerate's multi-GPU layer placement (`device_map="auto"`,
model_base.py:165-194) — sequential placement, not part of the fused-GEMV hot path — and the
hub download (`snapshot_download`; there is no network here, pass a local directory).
"""
from __future__ import annotations

import glob
import json
import os
from typing import Optional

import torch

from vptq_amd.layers.vqlinear import VQuantLinear, chain_prefetch, link_siblings


def make_quant_linear(module: torch.nn.Module, config_for_layers: dict, shared_layer_config: dict,
                      target_layer=VQuantLinear, dtype=None) -> int:
    """Replace, in place, every sub-module with a quantisation entry.  Returns the count."""
    n = 0
    for module_name, sub_module in list(module.named_modules()):
        tail = module_name.split(".")[-1]
        conf = config_for_layers.get(module_name)
        if conf is None and tail in shared_layer_config:
            conf = shared_layer_config[tail]
        if conf is None or not module_name:
            continue
        w = getattr(sub_module, "weight", None)
        kw = dict(conf)
        kw.setdefault("is_indice_packed", True)
        new = target_layer(**kw, enable_proxy_error=False,
                           dtype=dtype if dtype is not None else (w.dtype if w is not None else None))
        module.set_submodule(module_name, new)
        n += 1
    return n


def _checkpoint_files(path: str):
    idx = glob.glob(os.path.join(path, "*.safetensors.index.json"))
    if idx:
        with open(idx[0]) as f:
            files = sorted(set(json.load(f)["weight_map"].values()))
        return [os.path.join(path, f) for f in files]
    files = sorted(glob.glob(os.path.join(path, "*.safetensors")))
    if not files:
        raise FileNotFoundError(f"no *.safetensors under {path}")
    return files


_PLACEMENTS = ("auto", "balanced", "sequential", "balanced_low_0")


def _one_device(v) -> str:
    """HF's forms of "this device": an index (0, "0"), a torch.device, "cuda:1", "cpu" """
    if isinstance(v, torch.device):
        return str(v)
    if isinstance(v, bool):
        raise ValueError(f"device_map value {v!r} names no device")
    if isinstance(v, int) or (isinstance(v, str) and v.isdigit()):
        return f"cuda:{int(v)}"
    return str(v)


def _device_from_map(dm, device: Optional[str]) -> Optional[str]:
    """The ONE device a `device_map` asks for (an explicit `device` wins), None if it names none ("auto").
    accelerate's layer placement over several GPUs (reference vptq/layers/model_base.py:165-194) is not carried
    over: a map that asks for it is refused instead of being swallowed."""
    if dm is None:
        return device
    if isinstance(dm, dict):
        if len({_one_device(v) for v in dm.values()}) > 1:
            raise NotImplementedError("device_map places the model on several devices: this loader puts a model on ONE "
                                      "device (pass device=...); shard layers with vptq_amd.utils.shard for tensor parallelism")
        return device if (device is not None or not dm) else _one_device(next(iter(dm.values())))
    if isinstance(dm, str) and dm in _PLACEMENTS:
        if dm != "auto" and torch.cuda.device_count() > 1:
            raise NotImplementedError(f"device_map='{dm}' (multi-GPU placement by accelerate) is not supported: one device per model")
        return device   # (= the one device chosen by the caller: what accelerate does when the model fits one GPU)
    if isinstance(dm, (str, int, torch.device)) and not isinstance(dm, bool):
        return device if device is not None else _one_device(dm)   # ("cuda:1", "cpu", 1, "1", torch.device(...))
    raise TypeError(f"device_map of type {type(dm).__name__} is not understood")


class AutoModelForCausalLM:
    """`vptq.AutoModelForCausalLM` (reference vptq/__init__.py:7-14)."""

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path: str, *model_args,
                        device: Optional[str] = None, dtype: Optional[torch.dtype] = None,
                        link_prefetch: bool = False, absorb_perm: bool = False,
                        fuse_siblings: bool = False, **kwargs):
        import transformers
        from safetensors.torch import load_file

        device = _device_from_map(kwargs.pop("device_map", None), device)
        # ("auto" on this loader = the one device chosen below, what accelerate does when the model fits one GPU)
        if kwargs:
            import warnings
            warnings.warn(f"vptq_amd.AutoModelForCausalLM.from_pretrained ignores {sorted(kwargs)}", stacklevel=2)
        path = str(pretrained_model_name_or_path)
        if not os.path.isdir(path):
            raise FileNotFoundError(
                f"{path} is not a local directory (no hub access in this build: download the "
                "checkpoint and pass its path)")
        conf = transformers.AutoConfig.from_pretrained(path)
        qc = getattr(conf, "quantization_config", None)
        if qc is None:
            raise ValueError("config.json has no quantization_config: not a VPTQ checkpoint")
        qd = qc if isinstance(qc, dict) else qc.to_dict()
        if dtype is None:
            dtype = getattr(conf, "dtype", None) or getattr(conf, "torch_dtype", None) or torch.float16
        if isinstance(dtype, str):
            dtype = getattr(torch, dtype)
        if device is None:
            device = "cuda:0" if torch.cuda.is_available() else "cpu"

        # HF must not try to quantise by itself: we do the replacement below
        plain = conf.__class__.from_dict({k: v for k, v in conf.to_dict().items()
                                          if k != "quantization_config"})
        with torch.device("meta"):
            model = transformers.AutoModelForCausalLM.from_config(plain, *model_args, dtype=dtype)
            replaced = make_quant_linear(model, qd.get("config_for_layers", {}) or {},
                                         qd.get("shared_layer_config", {}) or {}, dtype=dtype)
        if replaced == 0:
            raise ValueError("quantization_config names no module of this model")
        model = model.to_empty(device=device)

        state = {}
        for f in _checkpoint_files(path):
            state.update(load_file(f, device=str(device)))
        missing, unexpected = model.load_state_dict(state, strict=False, assign=True)
        tied = getattr(plain, "tie_word_embeddings", False)
        missing = [k for k in missing if not (tied and k.startswith("lm_head."))]
        if missing or unexpected:
            raise RuntimeError(f"checkpoint does not match the model: missing {missing[:8]} "
                               f"unexpected {unexpected[:8]}")
        if tied:
            model.tie_weights()
        # non-persistent buffers (rotary inv_freq) were created on meta: rebuild them
        for name, mod in list(model.named_modules()):
            if type(mod).__name__.endswith("RotaryEmbedding"):
                model.set_submodule(name, type(mod)(config=plain).to(device))
        model.config.quantization_config = qd
        if absorb_perm:
            # fold every layer's input permutation into its index order (same dense weight,
            # no per-token activation gather; what the reference's tools/pre_process.py does)
            from vptq_amd.utils.pack import absorb_perm as _absorb
            _absorb(model)
        if fuse_siblings:
            # q/k/v and gate/up of a block read the same activation: one grouped launch each
            link_siblings(model)
        if link_prefetch:
            chain_prefetch([m for m in model.modules() if isinstance(m, VQuantLinear)], circular=True)
        return model.eval()
