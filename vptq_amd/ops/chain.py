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
            flags = B.GEMV_CHAIN_DEPENDENT if self.dependent else 0
            # Layers the load-time gate serves with the reference's roundings (VQuantLinear._folded_form_is_safe:
            # bias-dominated, fewer than 32 distinct vector-rows).  The arithmetic is a property of a LAUNCH, so
            # independent layers are handed to the library as two lists - the gated ones with VPTQ_GEMV_EXACT - and one
            # odd layer does not slow the others down; a dependent chain stays one list (the order is the
            # dependency) and takes the reference's roundings as a whole.
            gated = [i for i, c in enumerate(caches) if c[9] & B.GEMV_EXACT]
            rest = 0
            for c in caches:
                rest |= c[9] & B.GEMV_SELECTIVE      # (selective arithmetic: the un-gated layers' launch asks for it)
            if self.dependent or not gated or len(gated) == n:
                parts = [(list(range(n)), B.GEMV_EXACT if gated else rest)]
            else:
                parts = [([i for i in range(n) if not caches[i][9] & B.GEMV_EXACT], rest), (gated, B.GEMV_EXACT)]
            subs = []
            for idx, safe in parts:
                m = len(idx)
                descs = (B.LayerDesc * m)(*[caches[i][1] for i in idx])
                # arrival flags of a dependent chain / x[perm] of independent layers that have an input permutation / the
                # thresholds of a call with VPTQ_GEMV_SELECTIVE (one word per layer)
                nbytes = B.lib().vptq_quant_gemv_chain_workspace_bytes_for(descs, m, flags | (0 if self.dependent else B.GEMV_SELECTIVE))
                ws = torch.zeros(max(nbytes, 4) // 4, dtype=torch.int32, device=dev) if nbytes else None
                subs.append((idx, descs, (C.c_void_p * m)(), (C.c_void_p * m)(), safe, ws, nbytes))
            self._state = (key, subs, dev, caches[0][7], [c[2] for c in caches])
        return self._state

    def kernel_name(self, tokens: int = 1, flags: Optional[int] = None) -> Optional[str]:
        """what the library runs for the (first, i.e. un-gated) list of this chain"""
        _, subs, _, _, _ = self._prepare()
        idx, descs, _, _, safe, _, _ = subs[0]
        f = (ops.quant_gemm_flags() if flags is None else flags) | safe
        if self.dependent:
            f |= B.GEMV_CHAIN_DEPENDENT
        name = B.lib().vptq_quant_gemv_chain_kernel_name(descs, len(idx), tokens, f)
        return None if name is None else name.decode()

    def __call__(self, xs: Sequence[torch.Tensor], ys: Optional[Sequence[torch.Tensor]] = None,
                 flags: Optional[int] = None):
        """xs: one activation per layer (independent) or only the first layer's (dependent).
        ys: optional pre-allocated outputs.  Returns the list of outputs, one per layer."""
        _, subs, dev, wdtype, _ = self._prepare()
        n = len(self.layers)
        xs = list(xs)
        if self.dependent:
            if len(xs) != 1:
                raise ValueError("dependent chain: pass the first layer's activation only")
        elif len(xs) != n:
            raise ValueError(f"{n} layers need {n} activations")
        x0 = self.layers[0]._check_activation(xs[0])
        tokens = x0.numel() // x0.shape[-1]
        f0 = ops.quant_gemm_flags() if flags is None else flags
        out_dtype = torch.float32 if (f0 & B.GEMV_OUT_F32) else wdtype
        if self.dependent and (f0 & B.GEMV_OUT_F32):
            raise ValueError("a dependent chain feeds its outputs back in: no float32 outputs")
        if ys is None:
            ys = [torch.empty(x0.shape[:-1] + (m.out_features,), dtype=out_dtype, device=dev)
                  for m in self.layers]
        else:
            ys = list(ys)
            if len(ys) != n:
                raise ValueError(f"{n} layers need {n} outputs")
        keep = []
        xin = []
        for i, m in enumerate(self.layers):
            xi = ys[i - 1] if (self.dependent and i > 0) else m._check_activation(xs[i])
            if xi.numel() // xi.shape[-1] != tokens:
                raise ValueError("every layer of a chain takes the same number of tokens")
            if ys[i].dtype != out_dtype or ys[i].device != dev or not ys[i].is_contiguous() or \
                    ys[i].numel() != tokens * m.out_features:
                raise ValueError(f"output {i}: wrong dtype / device / size")
            keep.append(xi)
            xin.append(xi)
        if self.dependent:
            f0 |= B.GEMV_CHAIN_DEPENDENT
        with torch.cuda.device(dev):
            sp = B.current_stream_ptr(dev)
            for idx, descs, xp, yp, safe, ws, nbytes in subs:
                for j, i in enumerate(idx):
                    xp[j] = xin[i].data_ptr()
                    yp[j] = ys[i].data_ptr()
                rc = B.lib().vptq_quant_gemv_chain(descs, len(idx), xp, yp, tokens, f0 | safe,
                                                   None if ws is None else ws.data_ptr(), nbytes, sp)
                if rc:
                    B.check(rc, "vptq_quant_gemv_chain")
        self._keep = keep
        return ys


def quant_gemv_chain(layers, xs, dependent: bool = False):
    """One-shot form of `GemvChain` (builds the descriptor arrays on every call)."""
    return GemvChain(layers, dependent=dependent)(xs)
