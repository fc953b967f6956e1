"""`VQuantLinear`: drop-in for microsoft/VPTQ's quantised linear layer
(reference vptq/layers/vqlinear.py:17-240 ctor, :351-397 forward).

What must match the reference byte for byte is the *state-dict contract* HF
checkpoints rely on — parameter names, shapes and dtypes:

    centroids.weight       [C, k*v]            fp16/bf16
    res_centroids.weight   [C, kr*v]           (when num_res_centroids[1] > 0)
    indices                [C, N, ceil(G*T/32)] int32   (packed: idx | ridx << log2 k)
    outlier_centroids.weight [1, ko*ov], outlier_indices [1, M, S] int16
    perm                   [I] int16 (uint16 bit pattern)
    weight_scale, weight_bias [I];  bias [O]

and the constructor keyword arguments HF's `replace_with_vptq_linear` passes
(transformers/integrations/vptq.py).  The module may be built on the `meta`
device; nothing derived from tensor contents is computed in `__init__`.

Only the inference path is implemented: packed indices, `vector_quant_dim="out"`.
The reference's layer-wise fine-tuning helpers (proxy_error_forward,
set_l2_indices, init_parameters from k-means dictionaries) belong to the
quantisation algorithm, which is not part of this hot path.
"""
from __future__ import annotations

import math
from typing import Sequence

import os

import torch
import torch.nn as nn
from torch.nn.parameter import Parameter

from vptq_amd import _backend as B
from vptq_amd import ops


def chain_prefetch(layers, circular: bool = False):
    """Tell each layer which one runs next so that its fused GEMV reads the next layer's
    packed indices ahead into L2 / Infinity Cache (pure performance hint)."""
    layers = list(layers)
    for a, b in zip(layers, layers[1:] + ([layers[0]] if circular else [None])):
        object.__setattr__(a, "_prefetch_next", b)
    return layers


def _raw_stream(device_index: int) -> int:
    """hipStream_t of torch's current stream on `device_index` (no Stream object)."""
    return torch._C._cuda_getCurrentRawStream(device_index)


# VPTQ_SLICED_LAYOUT: "auto" (default) = every eligible large-codebook layer (v8-k65536-0 / -256) builds the sliced
# layout at its first one-token call while that leaves a quarter of the device memory free (MI355X: 288 GB - the
# layouts of a 70B 3-bit model are 45 GB); "1" = always; "0" = never (VQuantLinear.enable_sliced_layout per layer)
_SLICED_LAYOUT_MODE = os.environ.get("VPTQ_SLICED_LAYOUT", "auto").strip().lower() or "auto"
_SLICED_LAYOUT_ENV = _SLICED_LAYOUT_MODE not in ("0", "off", "false", "no")
_SLICED_MIN_FREE_FRACTION = 0.25
_SLICED_EXACT_MIN_ELEMENTS = 1 << 20   # vector-rows x columns from which the exact sliced kernel beats the gather kernel
_SLICED_EXACT_RG_MIN_ELEMENTS = 2 << 20   # ... with the residual entries gathered from L2 (two-table formats)
# most tokens served as one sliced launch PER TOKEN: decided per layer (VQuantLinear._sliced_token_limit);
# VPTQ_SLICED_TOKENS="one-table,two-table" overrides it (tools/sliced_tokens_bench.py)
_SLICED_TOKENS_ENV = tuple(int(v) for v in B.tune_env("VPTQ_SLICED_TOKENS").split(",")) if B.tune_env("VPTQ_SLICED_TOKENS") else None
_SLICED_MAX_TOKENS = max(max(_SLICED_TOKENS_ENV) if _SLICED_TOKENS_ENV else 3, 4)
# 2 - 4 tokens in ONE sliced launch (gemv_sliced_tok.hip): "auto" (default) = per layer where it was measured faster than the
# gather kernels and than one sliced launch per token (VQuantLinear._sliced_one_launch); "1" = wherever the library takes the
# layer; "0" = never
_SLICED_ONE_LAUNCH = B.tune_env("VPTQ_SLICED_ONE_LAUNCH", "auto").strip().lower() or "auto"


_SLICED_SELECTIVE_MIN_ELEMENTS = 6 << 20   # selective roundings over the folded sliced layouts (two-table formats): from 6 M index elements on
_SLICED_OOM_RETRY_CALLS = 256   # calls of a layer before a sliced-layout build that ran out of memory is tried again


class SiblingGroup:
    """Layers that are applied to the SAME activation one after the other (q / k / v, gate / up).
    The first member called with a tensor launches all members in one grouped kernel
    (`vptq_quant_gemv_grouped`: their workgroups share the launch, the prologue and the HBM
    stream); the others return their share of that launch when they are called with the very
    same tensor object (held alive, version checked), and fall back to their own launch
    otherwise.  Pure scheduling: the arithmetic per layer is that of a single launch."""
    MAX_TOKENS = 4

    def __init__(self, members):
        self.members = list(members)
        self._x = None
        self._version = -1
        self._out = {}
        self._arrays = None

    def forward_sliced(self, layer, x, tokens=1):
        """one token (2 - 4: `SlicedGroupGemv.forward_tokens`) of large-codebook siblings over their sliced layouts: ONE launch for the group
        (`vptq_amd/utils/sliced.py:SlicedGroupGemv`); same protocol as `forward` - the first member called launches, the
        others pick their output up when they are called with the very same tensor.  None = not this group's route (a
        member without a layout, mixed formats, a tensor without version counter, ...): the caller goes on alone."""
        ver = B.tensor_version(x)
        if ver >= 0 and self.__dict__.get("_sx") is x and ver == self._sversion and id(layer) in self._sout:
            y = self._sout.pop(id(layer))
            if not self._sout:
                self._sx = None
            return y
        if ver < 0:
            return None
        sls = [m._sliced_gemv() for m in self.members]
        if any(sl is None for sl in sls):
            return None
        key = tuple(id(sl) for sl in sls)
        sg = self.__dict__.get("_sgroup")
        if sg is None or sg[0] != key:
            from vptq_amd.utils.sliced import SlicedGroupGemv
            try:
                sg = (key, SlicedGroupGemv(sls))
            except ValueError:
                # mixed formats / arithmetics: every member launches for itself - remembered for THESE layouts only (the key):
                # a rebuilt layout or another arithmetic (set_arithmetic) makes new objects and the group is looked at again
                sg = (key, None)
            self._sgroup = sg
        if sg[1] is None:
            return None
        if tokens == 1:
            ys = sg[1](x)
        else:   # (every member must be a layer this route was measured faster for: VQuantLinear._sliced_one_launch)
            if not all(m._sliced_one_launch(sl, tokens) for m, sl in zip(self.members, sls)):
                return None
            ys = sg[1].forward_tokens(x)
        if ys is None:
            return None
        self._sx, self._sversion, self._keep_sx = x, ver, x
        self._sout = {id(m): y for m, y in zip(self.members, ys) if m is not layer}
        return ys[self.members.index(layer)]

    def forward(self, layer, x, tokens):
        # Tensors made under inference_mode track no version (-1): whether x was rewritten in place since the
        # leader's launch cannot be told then, so sibling outputs are not reused at all - every layer launches
        # for itself, as without a group.
        ver = B.tensor_version(x)
        if ver >= 0 and self._x is x and ver == self._version and id(layer) in self._out:
            y = self._out.pop(id(layer))
            if not self._out:
                self._x = None
            return y
        if ver < 0:
            self._x, self._out = None, {}
            return None
        self._out = {}     # a new leader call: whatever an earlier one left unconsumed is stale
        xc = layer._check_activation(x)
        caches = [m._descriptor() for m in self.members]
        dev = caches[0][3]
        if any(c[3] != dev for c in caches) or xc.device != dev or \
                any(m.in_features != layer.in_features for m in self.members):
            raise RuntimeError("sibling layers must share the device and the input width")
        key = tuple(c[6] for c in caches)  # descriptor generations: a rebuilt descriptor never matches
        if self._arrays is None or self._arrays[0] != key:
            import ctypes as C
            # one launch per ARITHMETIC: the members the load-time gate sends to the reference's roundings (VPTQ_GEMV_EXACT) go out as a
            # launch of their own, so that one such layer does not drag its siblings out of the selective / folded form (bf16 layers sit
            # close to the gate: with every group exact as soon as one member is, a decoder ran at the reference's speed)
            parts = []
            for want in (False, True):
                idx = [i for i, c in enumerate(caches) if bool(c[9] & B.GEMV_EXACT) == want]
                if idx:
                    fl = 0
                    for i in idx:
                        fl |= caches[i][9]
                    parts.append((idx, (B.LayerDesc * len(idx))(*[caches[i][1] for i in idx]), (C.c_void_p * len(idx))(),
                                  (C.c_void_p * len(idx))(), fl))
            self._arrays = (key, parts, B.lib().vptq_quant_gemv_grouped)
        _, parts, fn = self._arrays
        ys = [torch.empty(xc.shape[:-1] + (m.out_features,), dtype=xc.dtype, device=dev)
              for m in self.members]
        dev_index = caches[0][8]
        base_flags = ops.quant_gemm_flags()

        def launch(sp):
            for idx, descs, xp, yp, fl in parts:
                for j, i in enumerate(idx):
                    xp[j] = xc.data_ptr()
                    yp[j] = ys[i].data_ptr()
                rc = fn(descs, len(idx), xp, yp, tokens, base_flags | fl, sp)
                if rc:
                    B.check(rc, "vptq_quant_gemv_grouped")
        if torch.cuda.current_device() != dev_index:
            with torch.cuda.device(dev):
                launch(B.current_stream_ptr(dev))
        else:
            launch(_raw_stream(dev_index))
        self._x, self._version = x, B.tensor_version(x)
        self._keep_x = xc
        self._out = {id(m): y for m, y in zip(self.members, ys) if m is not layer}
        return ys[self.members.index(layer)]


SIBLING_PATTERNS = (("q_proj", "k_proj", "v_proj"), ("gate_proj", "up_proj"))


def link_siblings(model: nn.Module, patterns=SIBLING_PATTERNS) -> int:
    """Find, under every sub-module of `model`, children named like one of `patterns` that are all
    `VQuantLinear` layers of the same input width and dtype, and make each set a
    `SiblingGroup`.  Returns the number of groups.  Decode-time optimisation (1-4 tokens)."""
    n = 0
    for parent in model.modules():
        for names in patterns:
            kids = [getattr(parent, nm, None) for nm in names]
            if not all(isinstance(k, VQuantLinear) for k in kids):
                continue
            if len({(k.in_features, k.centroids.weight.dtype) for k in kids}) != 1:
                continue
            group = SiblingGroup(kids)
            for k in kids:
                object.__setattr__(k, "_siblings", group)
            n += 1
    return n


class VQuantLinear(nn.Module):
    _desc_generation = 0  # bumped for every descriptor built (SiblingGroup keys on it)

    def __init__(
        self,
        in_features: int,
        out_features: int,
        vector_lens: Sequence[int],
        num_centroids: Sequence[int],
        num_res_centroids: Sequence[int],
        group_num: int,
        group_size: int,
        outlier_size: int,
        indices_as_float: bool,
        enable_norm: bool = False,
        enable_perm: bool = False,
        is_indice_packed: bool = False,
        bias: bool = False,
        vector_quant_dim: str = "out",
        device=None,
        dtype=None,
        enable_proxy_error=True,
        **unused_kwargs,  # e.g. `norm_dim`, written by the reference's pack tools
    ):
        super().__init__()
        if vector_quant_dim not in ("in", "out"):
            raise ValueError("vector_quant_dim must be 'in' or 'out'.")
        if vector_quant_dim == "in":
            raise RuntimeError("Not implemented yet.")
        if not is_indice_packed:
            raise RuntimeError(
                "vptq_amd.VQuantLinear supports packed indices only (is_indice_packed=True): "
                "that is the format of every published VPTQ checkpoint; the reference's "
                "unpacked mode exists for fine-tuning.")
        fk = {"device": device, "dtype": dtype}
        self.vector_quant_dim = vector_quant_dim
        self.in_features, self.out_features = in_features, out_features
        self.enable_proxy_error = enable_proxy_error
        self.indices_as_float = indices_as_float
        self.is_indice_packed = True
        index_type = torch.float16 if indices_as_float else torch.int16

        if bias:
            self.bias = Parameter(torch.empty(out_features, **fk))
        else:
            self.register_parameter("bias", None)

        # main codebook(s): second element of the (outlier, main) pairs
        self.vector_len = vector_lens[1]
        self.num_centroids = num_centroids[1]
        self.group_num = self.num_codebooks = group_num
        self.group_size = group_size
        self.centroids = nn.Embedding(group_num, self.num_centroids * self.vector_len, **fk)

        # outlier columns: first element of the pairs
        self.outlier_size = outlier_size
        self.outlier_vector_len = vector_lens[0]
        self.num_outlier_centroids = num_centroids[0]
        self.outlier_num_res_centroids = num_res_centroids[0]
        self.enable_outlier = bool(self.outlier_vector_len > 1 and self.num_outlier_centroids > 0)
        self.outlier_padding = 0
        self.ouliter_num_indices = 0  # (sic) attribute name kept from the reference
        self.outlier_centroids = None
        self.outlier_indices = None
        if self.enable_outlier:
            if self.outlier_num_res_centroids != -1:
                raise ValueError("Current implementation does not support residual "
                                 "quantization on outliers yet.")
            self.outlier_padding = (-out_features) % self.outlier_vector_len
            self.ouliter_num_indices = (out_features + self.outlier_padding) // self.outlier_vector_len
            self.outlier_centroids = nn.Embedding(
                1, self.num_outlier_centroids * self.outlier_vector_len, **fk)
            self.outlier_indices = Parameter(
                torch.empty((1, self.ouliter_num_indices, outlier_size), dtype=index_type,
                            device=device), requires_grad=False)

        # residual codebook (indices ride inside the packed stream)
        self.num_res_centroids = num_res_centroids[1]
        self.enable_residual = self.num_res_centroids > 0
        self.res_indices = None
        if self.enable_residual:
            self.res_centroids = nn.Embedding(
                group_num, self.num_res_centroids * self.vector_len, **fk)
        else:
            self.register_parameter("res_centroids", None)

        self.enable_perm = enable_perm
        if enable_perm:
            self.perm = Parameter(torch.arange(in_features, device=device).to(torch.int16),
                                  requires_grad=False)

        self.enable_norm = enable_norm
        self.weight_scale = self.weight_bias = None
        if enable_norm:
            self.weight_scale = Parameter(torch.empty(in_features, **fk))
            self.weight_bias = Parameter(torch.empty(in_features, **fk))

        self.padding = (-out_features) % self.vector_len
        self.num_indices = (out_features + self.padding) // self.vector_len
        self.index_bits = int(math.log2(self.num_centroids))
        self.res_index_bits = int(math.log2(self.num_res_centroids)) if self.enable_residual else 0
        self.total_index_bits = self.index_bits + self.res_index_bits
        packed_groupsize = math.ceil(group_size * self.total_index_bits / 32)
        self.indices = Parameter(
            torch.empty((group_num, self.num_indices, packed_groupsize), dtype=torch.int32,
                        device=device), requires_grad=False)
        # optional: the layer that runs after this one (set by `chain_prefetch`); its packed
        # indices are read ahead by this layer's GEMV.  Not a parameter / buffer.
        self._prefetch_next = None

    def forward(self, x: torch.Tensor, W=None, H=None) -> torch.Tensor:
        """x [..., in_features] fp16/bf16 -> [..., out_features]."""
        if self.enable_proxy_error:
            raise RuntimeError(
                "enable_proxy_error=True selects the reference's layer-wise fine-tuning debug "
                "path, which is outside this inference package; construct the layer with "
                "enable_proxy_error=False (HF does).")
        tokens = x.numel() // x.shape[-1] if x.shape[-1] else 0
        if 1 <= tokens <= B.GEMV_MAX_TOKENS and x.is_cuda and \
                (tokens <= B.GEMV_ANY_FORMAT_TOKENS or tokens <= self._descriptor()[5]):
            return self._gemv_cached(x, tokens)
        if tokens >= 1 and x.is_cuda and ops.fused_gemm_max_tokens() < tokens:
            return self._dense_cached(x)
        return ops.quant_gemm(
            x,
            bias=self.bias,
            indices=self.indices,
            centroids=self.centroids.weight,
            outlier_indices=self.outlier_indices,
            outlier_centroids=self.outlier_centroids.weight if self.enable_outlier else None,
            residual_indices=None,
            residual_centroids=self.res_centroids.weight if self.enable_residual else None,
            perm=self.perm if self.enable_perm else None,
            weight_scale=self.weight_scale,
            weight_bias=self.weight_bias,
            vector_len=self.vector_len,
            outlier_vector_len=self.outlier_vector_len,
            num_codebooks=self.num_codebooks,
            num_centroids=self.num_centroids,
            num_outlier_centroids=self.num_outlier_centroids,
            num_res_centroids=self.num_res_centroids,
            is_indice_packed=True,
            group_size=self.group_size,
            outlier_size=self.outlier_size,
            in_features=self.in_features,
            out_features=self.out_features,
            padding=self.padding,
            outlier_padding=self.outlier_padding,
            vector_quant_dim=self.vector_quant_dim,
            prefetch=None if self._prefetch_next is None else self._prefetch_next.indices,
        )

    def _descriptor(self):
        """(desc, device, keep-alive): the C-ABI descriptor of this layer, built once and reused
        while the parameter storages stay the same (building it costs ~25 us of Python per call,
        several times the kernel itself)."""
        # parameters straight out of the module's dicts: nn.Module.__getattr__ costs ~0.4 us per name,
        # ten names per call were a third of this function (tools/py_overhead.py)
        P, M = self._parameters, self._modules
        nxt = self._prefetch_next
        # (an absent tensor is either a None entry of _parameters or a plain None attribute: .get covers both)
        perm = P.get("perm") if self.enable_perm else None
        tensors = (P["indices"], M["centroids"]._parameters["weight"],
                   M["res_centroids"]._parameters["weight"] if self.enable_residual else None,
                   P.get("outlier_indices"),
                   M["outlier_centroids"]._parameters["weight"] if self.enable_outlier else None,
                   perm, P.get("weight_scale"), P.get("weight_bias"),
                   P.get("bias"), None if nxt is None else nxt._parameters["indices"])
        # storage pointers of everything; version counters of the tensors the descriptor holds DERIVED
        # copies of (scale / bias in column order for `perm` layers), which an in-place update of the
        # parameters (load_state_dict's copy_, an optimizer step) must invalidate
        key = tuple(0 if t is None else t.data_ptr() for t in tensors) + (B.arithmetic_generation(),)
        if perm is not None:
            key += (B.tensor_version(perm), B.tensor_version(tensors[6]), B.tensor_version(tensors[7]))
        cache = self.__dict__.get("_desc_cache")
        if cache is None or cache[0] != key:
            dev = B.require_device(*[t for t in tensors if t is not None])
            desc, keep = B.make_layer_desc(
                indices=tensors[0], centroids=tensors[1], res_centroids=tensors[2],
                outlier_indices=tensors[3], outlier_centroids=tensors[4], perm=tensors[5],
                weight_scale=tensors[6], weight_bias=tensors[7], bias=tensors[8],
                in_features=self.in_features, out_features=self.out_features,
                vector_len=self.vector_len, num_codebooks=self.num_codebooks,
                num_centroids=self.num_centroids,
                num_res_centroids=self.num_res_centroids if self.enable_residual else 0,
                group_size=self.group_size,
                outlier_size=self.outlier_size if self.enable_outlier else 0,
                outlier_vector_len=self.outlier_vector_len,
                num_outlier_centroids=self.num_outlier_centroids, prefetch=tensors[9])
            VQuantLinear._desc_generation += 1
            cache = (key, desc, keep, dev, B.lib().vptq_quant_gemv,
                     B.lib().vptq_quant_gemv_max_tokens(desc), VQuantLinear._desc_generation,
                     tensors[1].dtype, dev.index if dev.index is not None else torch.cuda.current_device(),
                     B.layer_arithmetic_flags(self._folded_form_is_safe(tensors, desc)),
                     B.lib().vptq_quant_gemv_workspace_bytes(desc, 16, 0))   # [10]: scratch bytes of the batched-decode kernel
            self.__dict__["_desc_cache"] = cache
        return cache

    def _folded_form_is_safe(self, tensors, desc=None) -> bool:
        """Load-time gate of the library's default ("folded") decode arithmetic (`_backend.folded_form_is_safe`): the layer
        runs both forms on probe activations and keeps the folded one while their un-rounded outputs stay within
        `FOLDED_MAX_PROBE_DISTANCE` of max|y|; bias-dominated layers and layers with few distinct vector-rows get
        VPTQ_GEMV_EXACT (the reference's three roundings per weight).  One device -> host read per descriptor build, i.e.
        per layer load."""
        return B.folded_form_is_safe(tensors[0], tensors[1], tensors[2], tensors[6], tensors[7], desc,
                                     self.in_features, self.out_features)

    def _check_activation(self, x: torch.Tensor) -> torch.Tensor:
        if x.shape[-1] != self.in_features:
            raise RuntimeError(f"x has {x.shape[-1]} features, layer expects {self.in_features}")
        cw = self.centroids.weight
        if x.dtype != cw.dtype:
            raise RuntimeError(f"activation dtype {x.dtype} != weight dtype {cw.dtype}")
        if not x.is_cuda:
            raise RuntimeError("vptq_amd has no CPU path: x must be on the GPU")
        return x if x.is_contiguous() else x.contiguous()

    # derived, device-bound state (ctypes descriptors, the sliced layout, sibling links) is rebuilt on demand: it is
    # neither pickled nor deep-copied with the module (torch.save(model), copy.deepcopy(model))
    _DERIVED_STATE = ("_desc_cache", "_desc_dense", "_sliced", "_sliced_cand", "_siblings")

    def __getstate__(self):
        state = dict(self.__dict__)
        for k in self._DERIVED_STATE:
            state.pop(k, None)
        return state

    def __deepcopy__(self, memo):
        import copy
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k not in self._DERIVED_STATE:
                new.__dict__[k] = copy.deepcopy(v, memo)
        return new

    def enable_sliced_layout(self, enable: bool = True):
        """Opt this layer in to (out of) the load-time derived "sliced" layout of the large-codebook formats
        v8-k65536-0 / v8-k65536-256 (vptq_amd/utils/sliced.py, gemv_sliced.hip): one-token calls then run 2-2.8x
        faster (8192^2: 40 -> 14-19 us) for 1.7-2x the packed indices of extra device memory.  The layout is built
        at the first one-token call; the state-dict tensors are untouched.  Process-wide: VPTQ_SLICED_LAYOUT=1."""
        self.__dict__["_sliced_on"] = bool(enable)
        self.__dict__.pop("_sliced", None)

    def _sliced_gemv(self):
        on = self.__dict__.get("_sliced_on")
        if on is None:
            on = _SLICED_LAYOUT_ENV
        if not on:
            return None
        cache = self._descriptor()
        st = self.__dict__.get("_sliced")
        # (a rebuilt descriptor = other tensors; a bumped version counter = indices rewritten in place: rebuild)
        stamp = (cache[6], B.tensor_version(self._parameters["indices"]))
        if st is None or st[0] != stamp:
            if torch.cuda.is_current_stream_capturing():
                return None   # (no layout is built inside a capture - and the "no" is not remembered: a later call builds it)
            obj = None
            # the layer's arithmetic (cache[9]): the reference's roundings (the default) take the EXACT sliced kernel where it
            # serves the layer - no residual codebook or the 256-entry one of v = 8, up to ~16000 columns; the others keep the
            # gather kernel -, the opt-in folded form the folded one
            exact = bool(cache[9] & (B.GEMV_EXACT | B.GEMV_SELECTIVE))   # (selective: the one-table formats take the reference's roundings)
            # ... the two-table ones (v8-k65536-65536, v16-k65536-65536, ...: 59 / 46 us in the reference's roundings, the residual entry an
            # L2 gather per element) the FOLDED layouts with VPTQ_GEMV_SELECTIVE - a pre-pass zeroes the blocks an activation dominates and
            # hands their exact products to the folded launch (gemv_hot.hip): ~21 us
            kr0 = self.num_res_centroids if self.enable_residual else 0
            # (the pre-pass is a launch of its own, ~9 us at 8192^2: it pays on the large layers - from 6 M index elements on - where the
            # reference's roundings cost 46 - 59 us; smaller two-table layers keep the exact layouts)
            selective = bool(cache[9] & B.GEMV_SELECTIVE) and not (cache[9] & B.GEMV_EXACT) and kr0 >= 4096 and \
                (self.indices.shape[1] * self.group_size >= _SLICED_SELECTIVE_MIN_ELEMENTS or "_sliced_on" in self.__dict__) and \
                bool(B.lib().vptq_quant_gemv_sliced_selective_supported(cache[1])) and bool(B.lib().vptq_sliced_layout_supported_for(cache[1], 0))
            if selective:
                exact = False
            # (reference roundings, us per layer, gather -> exact sliced, profiles/r05/sliced_exact.txt: 8192^2 39.4 -> 17.2,
            # 4096 x 14336 34.5 -> 17.0, 14336 x 4096 34.9 -> 14.7, 4096^2 12.6 -> 9.0, 4096 x 1024 7.8 -> 8.3: from 1 M index elements -
            # 8 M weights - on; enable_sliced_layout() asks for it on any layer)
            n_el = self.indices.shape[1] * self.group_size
            kr_ = self.num_res_centroids if self.enable_residual else 0
            if exact and kr_ > 0 and not (self.vector_len == 8 and kr_ == 256):
                # any other residual codebook: its entries are gathered from L2 behind the LDS-local main gathers (2 blocks per
                # queue stage: the CU's L1 miss path is the limit) - us per layer, gather kernels -> sliced, profiles/r05/
                # sliced_exact_two_table_queue_ab.txt: v8-k65536-65536 8192^2 78.0 -> 59.9, 14336 x 4096 68.5 -> 52.2, 4096^2 21.6 ->
                # 21.2; v8-k65536-4096 69.9 -> 50.6 / 59.8 -> 44.3 / 20.5 -> 18.4; v16-k65536-65536 53.9 -> 46.7 / 52.3 -> 40.9 / 19.2 ->
                # 18.3; small residual tables (v16-k65536-1024: the gather kernel holds them in LDS) stay there
                big = (kr_ >= 4096 and n_el >= _SLICED_EXACT_RG_MIN_ELEMENTS) or "_sliced_on" in self.__dict__
            else:
                big = n_el >= _SLICED_EXACT_MIN_ELEMENTS or "_sliced_on" in self.__dict__
            from vptq_amd.utils.sliced import SlicedGemv, exact_column_parts
            # (reference roundings: layers too wide for the LDS in one piece - 28672 columns - are served as equal column parts)
            served = exact_column_parts(cache[1], self.group_size)[0] if exact else B.lib().vptq_sliced_layout_supported_for(cache[1], 0)
            if selective:
                big = True
            if (big or not (exact or selective)) and served and self._sliced_fits(cache, on):
                # (a build that ran out of device memory is retried only after a back-off: every attempt costs int64 / float64
                # temporaries of ~160 bytes per element, an empty_cache() and a warning - per decode call, while memory stays tight)
                oom = self.__dict__.get("_sliced_oom")
                if oom is not None:
                    oom[0] -= 1
                    free = torch.cuda.mem_get_info(cache[3])[0] if oom[0] > 0 else 0
                    if oom[0] > 0 and free < oom[1]:
                        return None
                try:
                    obj = SlicedGemv(self, exact=exact, selective=selective)
                    self.__dict__.pop("_sliced_oom", None)
                except torch.cuda.OutOfMemoryError as e:
                    # out of device memory while building: the regular route serves the calls until _SLICED_OOM_RETRY_CALLS further
                    # ones have passed or the device reports twice the free bytes it had now.  Any other error is a bug and propagates.
                    import warnings
                    torch.cuda.empty_cache()
                    if oom is None:   # (warn once per layer)
                        warnings.warn(f"sliced layout of a {self.in_features} x {self.out_features} layer not built "
                                      f"({str(e)[:120]}); the layer keeps the gather kernel for now", stacklevel=3)
                    self.__dict__["_sliced_oom"] = [_SLICED_OOM_RETRY_CALLS, 2 * torch.cuda.mem_get_info(cache[3])[0] + (n_el * 160)]
                    return None
                # the kernel over the layouts evaluates the folded form: the same measured gate as every folded route
                # (_backend.folded_form_is_safe) - its float32 outputs against the gather kernel's (the reference's roundings)
                # on the probe activations
                lim = None if exact else (B.SELECTIVE_MAX_PROBE_DISTANCE if selective else B.FOLDED_MAX_PROBE_DISTANCE).get(cache[7])
                if lim is not None and self._parameters.get("weight_bias") is not None:
                    run = obj

                    def folded(xr, yr):
                        return run(xr.view(1, 1, -1), yr.view(1, 1, -1), flags=B.GEMV_OUT_F32) is not None
                    d = B.folded_probe_distance(cache[1], self.in_features, self.out_features, self._parameters["weight_bias"],
                                                cache[7], cache[3], folded=folded)
                    if not bool((d <= lim).item()):
                        obj = None
                        if selective:
                            # the gate refused the selective form of this layer: the reference's roundings over an exact layout
                            # (main entry from LDS, residual entry from L2), where that route serves it
                            if exact_column_parts(cache[1], self.group_size)[0] and self._sliced_fits(cache, on):
                                try:
                                    obj = SlicedGemv(self, exact=True)
                                except (torch.cuda.OutOfMemoryError, ValueError):
                                    obj = None
            st = (stamp, obj)
            self.__dict__["_sliced"] = st
        return st[1]

    def _sliced_token_limit(self, sl) -> int:
        """most tokens served as one sliced launch per token (measured, profiles/r04/sliced_tokens.txt: 8192^2 / 4096 x 14336 /
        4096^2, us per layer, gather kernel against T sliced launches - v8-k65536-0: 2 tokens 39.6 / 36.2 / 13.2 against 26.4 /
        27.1 / 17.0; v8-k65536-256: 41.4 / 36.7 / 13.4 against 34.0 / 34.5 / 19.5; v8-k65536-65536: 2 tokens 78.6 / 66.6 / 22.0
        against 42.1 / 41.3 / 23.5, 3 tokens 80.9 / 69.1 against 62.2 / 61.7; v16-k65536-65536: 2 tokens 52.9 / 52.0 / 19.6
        against 40.2 / 44.0 / 24.1; small residual tables of v = 16: never)"""
        lim = sl.__dict__.get("_token_limit")
        if lim is None:
            if sl.exact:
                # the reference's roundings: the gather kernels take 2 - 8 tokens for the price of one; TWO exact sliced launches beat
                # them on large v = 8 one-table layers only (profiles/r05/sliced_tokens_exact.txt, gather -> 2 launches, v8-k65536-256 /
                # -0: 8192^2 41.8 -> 40.0 / 41.4 -> 32.9; 14336 x 4096 37.0 -> 32.6 / 37.1 -> 27.9; 8192 x 28672 148.7 -> 94.8 / 135.1 ->
                # 80.9; but 4096 x 14336 36.7 -> 37.4, 4096^2 13.5 -> 19.5)
                n_el = self.indices.shape[1] * self.group_size
                kr = self.num_res_centroids if self.enable_residual else 0
                lim = 2 if (self.vector_len == 8 and kr in (0, 256) and sl.slices >= 16 and n_el >= 6 << 20) else 1
            elif _SLICED_TOKENS_ENV is not None:
                lim = _SLICED_TOKENS_ENV[1 if len(sl.layout) == 2 else 0]
            else:
                n_el = self.indices.shape[1] * self.group_size      # elements per table
                kr = self.num_res_centroids if self.enable_residual else 0
                lim = 1
                if self.vector_len == 8 and kr in (0, 256) and n_el >= 6 << 20:
                    lim = 2
                elif kr >= 16384 and n_el >= 3 << 20:
                    lim = 3 if (self.vector_len == 8 and kr == 65536 and n_el >= 6 << 20) else 2
            sl.__dict__["_token_limit"] = lim
        return lim

    def _sliced_one_launch(self, sl, tokens: int) -> bool:
        """2 - 4 tokens in ONE launch over the layouts (column phases; 3 - 4 tokens - in the reference's roundings: 2 - 4 - the
        contraction on the matrix pipe - gemv_sliced_tok.hip)?  Measured against the gather kernels (profiles/r04/sliced_tokens_one_launch.txt; us per 8192^2 /
        4096^2 / 14336 x 4096 layer, gather -> one launch): v8-k65536-0: 2 tokens 40.8 / 12.9 / 36.6 -> 21.4 / 11.5 / 19.0, 4 tokens
        41.1 / 13.4 / 36.9 -> 26.0 / 13.5 / 22.8; -256: 40.9 / 13.2 / 36.8 -> 23.7 / 12.4 / 20.4 and 44.0 / 16.0 / 42.9 -> 29.4 / 14.8 / 23.9;
        -65536: 78.2 / 21.5 / 68.8 -> 33.7 / 15.9 / 28.5 and 80.0 / 23.6 / 72.0 -> 39.8 / 18.5 / 43.6; v16-k65536-0: 27.7 / 13.7 / 33.5 ->
        23.4 / 12.8 / 19.5 and 31.1 / 15.2 -> 25.6 / 14.4; v16-k65536-65536: 53.0 / 19.6 / 53.9 -> 37.7 / 17.8 / 36.1 and 55.8 / 21.0 ->
        46.0 / 20.5 (4096 x 14336: 7 rows per wave - their sums in LDS, two rounds of workgroups: 57.2 -> 58.0, not taken).
        v = 16 with a residual table of <= 1024 entries stays on the gather kernel, which holds that table in LDS (v16-k65536-1024,
        2 tokens: 28.3 -> 36.1)."""
        key = ("_one_launch", tokens)
        ok = sl.__dict__.get(key)
        if ok is None:
            if _SLICED_ONE_LAUNCH in ("0", "off", "false", "no") or not sl.tokens_supported(tokens):
                ok = False
            elif _SLICED_ONE_LAUNCH in ("1", "on", "true", "yes", "always"):
                ok = True
            elif sl.exact:
                # the reference's roundings (gemv_sliced_tok.hip, EX; profiles/r05/sliced_tokens_exact.txt, gather -> one launch, us per
                # layer, v8-k65536-256): 8192^2: 2 / 3 / 4 tokens 41.5 / 43.7 / 42.0 -> 37.5 / 36.0 / 36.8; 14336 x 4096: 36.7 / 41.0 / 41.3 ->
                # 37.4 / 37.5 / 38.0; 8192 x 28672: 148.7 / 146.6 / 147.9 -> 122.0 / 124.5 / 127.0 (-0: 135 -> 82 - 86); but 4096^2: 13.5 / 15.8 / 15.7 -> 20.9 / 20.7 / 20.8 and 4096 x 14336: 36.7 / 36.8 / 37.0 -> 41.1 / 41.3 /
                # 41.9 (8 slices of 128 KiB leave room for a quarter of the columns: 4 phases); v = 16: 28.5 -> 36.0.  5 - 8 tokens: never
                # (8192^2: 44 - 46 -> 54 - 55)
                # 2 tokens there: two launches of the one-token kernel are as fast or faster (_sliced_token_limit).
                # 2 / 3 tokens of layers whose slice leaves room for (2 tokens + 4) bytes per column - 16-slice layouts up to ~12000 /
                # ~9700 columns - take ONE PASS of the one-token kernel (gemv_sliced.hip, TOK; profiles/r05/sliced_exact_tokens_one_pass.txt,
                # gather -> one pass, 2 / 3 tokens): 8192^2 42.1 / 44.9 -> 25.1 / 28.3; 8192 x 28672 149 / 147 -> 62 / 76; 8192 x 1024 13.0 / 18.0
                # -> 11.0 / 12.9 (k65536-0: 41.3 / 42.2 -> 22.2 / 25.9, 134 / 135 -> 53 / 66); narrow layers that fit with 8 slices: parity
                # (2048 x 8192: 13.5 / 16.0 -> 13.5 / 15.5), v = 16: slower (8192^2 28.6 -> 30.3) - both stay on the gather kernel
                # two tables (v8-k65536-65536, -4096: the residual entries gathered from L2 ONCE for all tokens; profiles/r05/
                # sliced_exact_tokens_one_pass.txt): 8192^2 77.8 -> 61.5 / 63.4, 8192 x 28672 255 -> 202 / 206, kr = 4096: 68 -> 53 / 54; small
                # layers lose (8192 x 1024: 18.8 -> 21.1): the one-token rule's sizes
                n_el = self.indices.shape[1] * self.group_size
                kr = self.num_res_centroids if self.enable_residual else 0
                # ... in WINDOW PARTS where only half of the columns' operands fit beside the slice (WPT; profiles/r05/
                # sliced_exact_tokens_window_parts.txt, gather -> one pass, 2 / 3 tokens, k65536-256 / -0): 14336 x 4096 37.0 / 42.6 -> 24.9 / 28.4
                # and 36.9 / 35.3 -> 21.7 / 26.2; 4096 x 14336 37.1 / 38.2 -> 31.5 / 37.1 and 36.2 / 35.0 -> 27.9 / 34.2; smaller layers lose
                # (4096^2 13.6 / 16.2 -> 14.8 / 16.7): from 6 M index elements on
                if self.vector_len != 8:
                    ok = False
                elif kr not in (0, 256):
                    ok = sl.slices >= 16 and tokens <= 3 and kr >= 4096 and n_el >= _SLICED_EXACT_RG_MIN_ELEMENTS and sl.tokens_one_pass(tokens)
                else:
                    wparts = sl.tokens_window_parts(tokens) if tokens <= 3 else 0
                    if wparts == 1:
                        ok = sl.slices >= 16
                    elif wparts > 1:
                        ok = n_el >= 6 << 20
                    else:
                        ok = sl.slices >= 16 and 3 <= tokens <= 4 and n_el >= 6 << 20
            else:
                n_el = self.indices.shape[1] * self.group_size      # index elements per table
                kr = self.num_res_centroids if self.enable_residual else 0
                if n_el < 1 << 19:       # (smaller layers are launch-bound on every route and were not measured)
                    ok = False
                elif self.vector_len == 8 or kr == 0:
                    ok = True
                elif kr <= 1024:
                    ok = False
                else:
                    ok = tokens == 2 or self.out_features <= 8192
            sl.__dict__[key] = ok
        return ok

    def _sliced_fits(self, cache, on) -> bool:
        """auto mode: build only while the layout (5 / 4 bytes per element + the builder's temporaries) leaves
        _SLICED_MIN_FREE_FRACTION of the device memory free, and never inside a stream capture"""
        if torch.cuda.is_current_stream_capturing():
            return False
        if on is True and "_sliced_on" in self.__dict__ or _SLICED_LAYOUT_MODE in ("1", "on", "true", "yes", "always"):
            return True
        free, total = torch.cuda.mem_get_info(cache[3])
        elems = self.indices.shape[1] * self.group_size
        two = B.lib().vptq_sliced_layout_tables(cache[1]) == 2   # (one layout per table)
        need = elems * (8 if two else 5) + elems * 8 * 20   # layout + int64 / float64 temporaries of build_sliced_layout (window order: more of them)
        return free - need > _SLICED_MIN_FREE_FRACTION * total

    def _gemv_cached(self, x: torch.Tensor, tokens: int) -> torch.Tensor:
        """Decode fast path: identical to `ops.quant_gemm` for 1..8 (canonical format: 16) tokens with a cached
        descriptor; layers linked by `link_siblings` share one grouped launch."""
        if tokens <= _SLICED_MAX_TOKENS and self.__dict__.get("_sliced_cand", True) and (_SLICED_LAYOUT_ENV or "_sliced_on" in self.__dict__):
            if "_sliced_cand" not in self.__dict__:
                # (static module configuration: decided once, so that every other layer pays one dict look-up per call)
                self.__dict__["_sliced_cand"] = bool(
                    self.num_centroids >= 16384 and self.vector_len in (8, 16) and self.num_codebooks == 1 and
                    not self.enable_outlier and self.enable_norm)   # (a permutation may still be absorbed later; the library
                    # decides: vptq_sliced_layout_supported)
            sl = self._sliced_gemv() if self.__dict__["_sliced_cand"] else None
            gf = ops.quant_gemm_flags()
            if sl is not None and not (gf & B.GEMV_FORCE_GENERIC) and (sl.exact or not (gf & B.GEMV_EXACT)):
                if tokens == 1:
                    sib = self.__dict__.get("_siblings")
                    if sib is not None:      # q / k / v, gate / up: one sliced launch for the group
                        y = sib.forward_sliced(self, x)
                        if y is not None:
                            return y
                    y = sl(x)
                    if y is not None:   # (None: misaligned activation, capture on a stream the layer has not run on, ...)
                        return y
                elif tokens <= 4 and self._sliced_one_launch(sl, tokens) and x.is_contiguous():
                    sib = self.__dict__.get("_siblings")
                    if sib is not None:      # q / k / v, gate / up: one launch for the group
                        y = sib.forward_sliced(self, x, tokens)
                        if y is not None:
                            return y
                    y = sl.forward_tokens(x)
                    if y is not None:   # (None: misaligned activation, capture on a stream without a workspace yet)
                        return y
                elif tokens <= self._sliced_token_limit(sl) and x.is_contiguous() and (self.in_features * x.element_size()) % 16 == 0:
                    # 2 (the 4-bit format: 3) tokens of a LARGE layer = one launch per token over the layouts: the gather
                    # kernels cost about as much for one token as for four (they are bound by the table gathers, not by
                    # the FMAs), the sliced kernel a third to a half of that per token (profiles/r04/sliced_tokens.txt)
                    x2 = x.reshape(tokens, self.in_features)
                    y = torch.empty(x.shape[:-1] + (self.out_features,), dtype=x.dtype, device=x.device)
                    y2 = y.view(tokens, self.out_features)
                    ok = True
                    for t in range(tokens):
                        if sl(x2[t], y2[t]) is None:
                            ok = False
                            break
                    if ok:
                        return y
        group = self.__dict__.get("_siblings")
        if group is not None and tokens <= group.MAX_TOKENS:
            y = group.forward(self, x, tokens)
            if y is not None:
                return y
        _, desc, _, dev, fn, _, _, wdtype, dev_index, safe_flags, ws_bytes = self._descriptor()
        # (the checks of _check_activation against the cached dtype / device: no module attribute look-ups)
        if x.shape[-1] != self.in_features:
            raise RuntimeError(f"x has {x.shape[-1]} features, layer expects {self.in_features}")
        if x.dtype != wdtype:
            raise RuntimeError(f"activation dtype {x.dtype} != weight dtype {wdtype}")
        if x.device != dev:
            if not x.is_cuda:
                raise RuntimeError("vptq_amd has no CPU path: x must be on the GPU")
            raise RuntimeError(f"tensors on different devices: {dev} vs {x.device}")
        if not x.is_contiguous():
            x = x.contiguous()
        y = torch.empty(x.shape[:-1] + (self.out_features,), dtype=wdtype, device=dev)
        # the current stream of the layer's device as a raw handle (torch.cuda.current_stream builds a
        # Stream object per call: 4 us of the 15 this function took)
        if torch.cuda.current_device() != dev_index:
            with torch.cuda.device(dev):
                sp = B.current_stream_ptr(dev)
                ws, wsb = B.gemv_workspace(dev_index, sp, ws_bytes) if tokens > 1 else (None, 0)
                rc = fn(desc, x.data_ptr(), y.data_ptr(), tokens, ops.quant_gemm_flags() | safe_flags, ws, wsb, sp)
        else:
            sp = _raw_stream(dev_index)
            # 2+ tokens of the canonical format: the one-pass batched-decode kernel wants scratch memory
            ws, wsb = B.gemv_workspace(dev_index, sp, ws_bytes) if tokens > 1 else (None, 0)
            rc = fn(desc, x.data_ptr(), y.data_ptr(), tokens, ops.quant_gemm_flags() | safe_flags, ws, wsb, sp)
        if rc == -5 and tokens > B.GEMV_ANY_FORMAT_TOKENS:
            # VPTQ_E_TOKENS: the fused path takes this layer's 17+ tokens only under run-time conditions the
            # descriptor cannot promise (16-byte aligned activations, no FORCE_* flag): the dense route
            return self._dense_cached(x)
        if rc:
            B.check(rc, "vptq_quant_gemv")
        return y

    def _dense_cached(self, x: torch.Tensor) -> torch.Tensor:
        """Many tokens: `vptq_dequant` into a fresh dense W + `F.linear` (the reference's route,
        vptq/ops/quant_gemm.py:231-274) - what `ops.quant_gemm` does, with the descriptor (and its
        argsort(perm)) cached: marshalling 28 keyword arguments and rebuilding the descriptor cost
        ~45 us of Python per layer, a third of the prompt pass of an 8B-shaped model at 128 tokens."""
        cache = self._descriptor()
        dev, wdtype, dev_index = cache[3], cache[7], cache[8]
        if x.shape[-1] != self.in_features:
            raise RuntimeError(f"x has {x.shape[-1]} features, layer expects {self.in_features}")
        if x.dtype != wdtype:
            raise RuntimeError(f"activation dtype {x.dtype} != weight dtype {wdtype}")
        if x.device != dev:
            raise RuntimeError(f"tensors on different devices: {dev} vs {x.device}")
        dense = self.__dict__.get("_desc_dense")
        if dense is None or dense[0] != cache[6]:
            P, M = self._parameters, self._modules
            desc, keep = B.make_layer_desc(
                indices=P["indices"], centroids=M["centroids"]._parameters["weight"],
                res_centroids=M["res_centroids"]._parameters["weight"] if self.enable_residual else None,
                outlier_indices=P.get("outlier_indices"),
                outlier_centroids=M["outlier_centroids"]._parameters["weight"] if self.enable_outlier else None,
                perm=P.get("perm") if self.enable_perm else None,
                weight_scale=P.get("weight_scale"), weight_bias=P.get("weight_bias"), bias=None,
                in_features=self.in_features, out_features=self.out_features,
                vector_len=self.vector_len, num_codebooks=self.num_codebooks,
                num_centroids=self.num_centroids,
                num_res_centroids=self.num_res_centroids if self.enable_residual else 0,
                group_size=self.group_size,
                outlier_size=self.outlier_size if self.enable_outlier else 0,
                outlier_vector_len=self.outlier_vector_len,
                num_outlier_centroids=self.num_outlier_centroids, need_inv_perm=True)
            dense = (cache[6], desc, keep, B.lib().vptq_dequant)
            self.__dict__["_desc_dense"] = dense
        W = torch.empty((self.out_features, self.in_features), dtype=wdtype, device=dev)
        if torch.cuda.current_device() != dev_index:
            with torch.cuda.device(dev):
                rc = dense[3](dense[1], W.data_ptr(), B.current_stream_ptr(dev))
        else:
            rc = dense[3](dense[1], W.data_ptr(), _raw_stream(dev_index))
        if rc:
            B.check(rc, "vptq_dequant")
        return torch.nn.functional.linear(x, W, self._parameters.get("bias"))

    def dequant(self) -> torch.Tensor:
        """Dense W[out_features, in_features] (what the reference calls
        `ops.dequant(...)` with this layer's fields)."""
        return ops.dequant(
            indices=self.indices, centroids=self.centroids.weight,
            outlier_indices=self.outlier_indices,
            outlier_centroids=self.outlier_centroids.weight if self.enable_outlier else None,
            res_indices=None,
            res_centroids=self.res_centroids.weight if self.enable_residual else None,
            perm=self.perm if self.enable_perm else None, weight_scale=self.weight_scale,
            weight_bias=self.weight_bias, is_indice_packed=True,
            enable_outlier=self.enable_outlier, enable_residual=self.enable_residual,
            enable_perm=self.enable_perm, enable_norm=self.enable_norm,
            num_centroids=self.num_centroids, num_outlier_centroids=self.num_outlier_centroids,
            num_res_centroids=self.num_res_centroids, padding=self.padding,
            outlier_padding=self.outlier_padding, num_codebooks=self.num_codebooks,
            group_size=self.group_size, outlier_size=self.outlier_size,
            vector_len=self.vector_len, outlier_vector_len=self.outlier_vector_len)

    def extra_repr(self) -> str:
        return (f"in_features={self.in_features}, out_features={self.out_features}, "
                f"v={self.vector_len}, k={self.num_centroids}, k_res={self.num_res_centroids}, "
                f"codebooks={self.num_codebooks}, group_size={self.group_size}, "
                f"outlier_size={self.outlier_size}, perm={self.enable_perm}, "
                f"norm={self.enable_norm}")
