"""Several `VQuantLinear` GEMVs in ONE persistent launch (`vptq_quant_gemv_chain`, ABI 6).

The reference runs a decode step as one `quant_gemv` call per layer
(vptq/ops/quant_gemm.py:214-228): every call pays a launch boundary, the codebooks' way into
shared memory and an epilogue.  `GemvChain` hands the library a LIST of layers; for the canonical
v=8 / 256+256 format it becomes one launch whose workgroups walk the layers one after the other
and request layer i + 1's codebooks, activations and first index words while layer i streams
(`vptq_amd/csrc/gemv_k256c.hip`); anything else is executed layer by layer by the library.
Pure scheduling: per layer the results are those of `VQuantLinear.forward`.

    chain = GemvChain([q_proj, k_proj, v_proj])          # independent layers
    yq, yk, yv = chain([x, x, x])
    chain = GemvChain(layers, dependent=True)            # x of layer i + 1 is y of layer i
    ys = chain([x0])                                     # ys[-1] = the chain's output
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import torch

from vptq_amd import _backend as B
from vptq_amd import ops


class GemvChain:
    def __init__(self, layers: Sequence, dependent: bool = False):
        self.layers = list(layers)
        if not 1 <= len(self.layers) <= B.CHAIN_MAX:
            raise ValueError(f"a chain holds 1..{B.CHAIN_MAX} layers")
        self.dependent = bool(dependent)
        if self.dependent:
            for a, b in zip(self.layers, self.layers[1:]):
                if a.out_features != b.in_features:
                    raise ValueError("dependent chain: out_features of a layer must be in_features of the next")
        self._state = None

    def _prepare(self):
        caches = [m._descriptor() for m in self.layers]
        key = tuple(c[6] for c in caches)
        if self._state is None or self._state[0] != key:
            n = len(caches)
            dev = caches[0][3]
            if any(c[3] != dev for c in caches):
                raise RuntimeError("the layers of a chain must share one device")
            descs = (B.LayerDesc * n)(*[c[1] for c in caches])
            flags = B.GEMV_CHAIN_DEPENDENT if self.dependent else 0
            nbytes = B.lib().vptq_quant_gemv_chain_workspace_bytes(n, flags)
            ws = torch.zeros(max(nbytes, 4) // 4, dtype=torch.int32, device=dev) if nbytes else None
            safe = 0
            for c in caches:
                safe |= c[9]   # a bias-dominated layer (VQuantLinear._folded_form_is_safe): reference arithmetic
            self._safe_flags = safe
            self._state = (key, descs, (C.c_void_p * n)(), (C.c_void_p * n)(), ws, nbytes, dev,
                           caches[0][7], [c[2] for c in caches])
        return self._state

    def kernel_name(self, tokens: int = 1, flags: Optional[int] = None) -> Optional[str]:
        _, descs, _, _, _, _, _, _, _ = self._prepare()
        f = (ops.quant_gemm_flags() if flags is None else flags) | self._safe_flags
        if self.dependent:
            f |= B.GEMV_CHAIN_DEPENDENT
        name = B.lib().vptq_quant_gemv_chain_kernel_name(descs, len(self.layers), tokens, f)
        return None if name is None else name.decode()

    def __call__(self, xs: Sequence[torch.Tensor], ys: Optional[Sequence[torch.Tensor]] = None,
                 flags: Optional[int] = None):
        """xs: one activation per layer (independent) or only the first layer's (dependent).
        ys: optional pre-allocated outputs.  Returns the list of outputs, one per layer."""
        _, descs, xp, yp, ws, nbytes, dev, wdtype, _ = self._prepare()
        n = len(self.layers)
        xs = list(xs)
        if self.dependent:
            if len(xs) != 1:
                raise ValueError("dependent chain: pass the first layer's activation only")
        elif len(xs) != n:
            raise ValueError(f"{n} layers need {n} activations")
        x0 = self.layers[0]._check_activation(xs[0])
        tokens = x0.numel() // x0.shape[-1]
        f = (ops.quant_gemm_flags() if flags is None else flags) | self._safe_flags
        out_dtype = torch.float32 if (f & B.GEMV_OUT_F32) else wdtype
        if self.dependent and (f & B.GEMV_OUT_F32):
            raise ValueError("a dependent chain feeds its outputs back in: no float32 outputs")
        if ys is None:
            ys = [torch.empty(x0.shape[:-1] + (m.out_features,), dtype=out_dtype, device=dev)
                  for m in self.layers]
        else:
            ys = list(ys)
            if len(ys) != n:
                raise ValueError(f"{n} layers need {n} outputs")
        keep = []
        for i, m in enumerate(self.layers):
            xi = ys[i - 1] if (self.dependent and i > 0) else m._check_activation(xs[i])
            if xi.numel() // xi.shape[-1] != tokens:
                raise ValueError("every layer of a chain takes the same number of tokens")
            if ys[i].dtype != out_dtype or ys[i].device != dev or not ys[i].is_contiguous() or \
                    ys[i].numel() != tokens * m.out_features:
                raise ValueError(f"output {i}: wrong dtype / device / size")
            keep.append(xi)
            xp[i] = xi.data_ptr()
            yp[i] = ys[i].data_ptr()
        if self.dependent:
            f |= B.GEMV_CHAIN_DEPENDENT
        with torch.cuda.device(dev):
            rc = B.lib().vptq_quant_gemv_chain(descs, n, xp, yp, tokens, f,
                                               None if ws is None else ws.data_ptr(), nbytes,
                                               B.current_stream_ptr(dev))
        if rc:
            B.check(rc, "vptq_quant_gemv_chain")
        self._keep = keep
        return ys


def quant_gemv_chain(layers, xs, dependent: bool = False):
    """One-shot form of `GemvChain` (builds the descriptor arrays on every call)."""
    return GemvChain(layers, dependent=dependent)(xs)
