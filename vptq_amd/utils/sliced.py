"""Load-time derived layout for large-codebook layers (v = 8, k = 65536, residual none or 256: "v8-k65536-0",
"v8-k65536-256"), the formats of most published VPTQ checkpoints, and the one-token GEMV over it
(`vptq_quant_gemv_sliced`, vptq_amd/csrc/gemv_sliced.hip).

The reference gathers centroid rows from the 1 MiB codebook through the caches for every index
(csrc/kernels/quant_gemv.cuh:11-186); on MI355X that is bound by the L2 -> L1 fill rate (39.6 us per 8192^2
layer).  Bucketing every row's elements ONCE by the top 3 (wide layers: 4) bits of their index
lets a workgroup keep its 8192- (4096-) entry slice of the codebook in LDS.  The state-dict tensors are untouched (they stay
the contract, and the many-token / dequant paths keep using them); the derived tensors cost 2.03x (T = 16) / 1.36x
(T = 24: 4 instead of 3 bytes per element + 8 bytes per block of <= 64 elements) the packed indices in device memory
on top.

    sl = SlicedGemv(layer)          # builds the layout (torch, on the layer's device)
    y = sl(x)                       # one token; same result as layer(x) within the parity bar
"""
from __future__ import annotations

import ctypes as C

import torch

from vptq_amd import _backend as B

INDEX_BITS = 16   # 65536 main centroids


CELL_BITS = 11   # a block's elements share a cell of 2048 columns: 11-bit column offsets


def build_sliced_layout(indices: torch.Tensor, group_size: int, slices: int = 8, residual: bool = False):
    """indices: the layer's packed int32 `indices` [1, N, row_words]: a little-endian bit stream per row, element g
    at bits [T g, T g + T) with value (residual index << 16) | main index (vptq/utils/pack.py:26-89); T = 16 without
    a residual codebook, 24 with 256 residual centroids.  slices: 8 or 16 (vptq_sliced_layout_supported tells).
    Returns the tensors of include/vptq_hip.h:VptqSlicedLayout:
      elems  int32 [elements + 64]  element words in (slice, row, cell, column) order, no padding between lists:
                                    column offset in its 2048-column cell | local index << 11 | residual index << 24
      bstart int32 [blocks]         first element of the block;  bmeta int32 [blocks] = valid elements (1..64) | cell << 8
      blocks int32 [slices, N]      blocks of (slice, row);  first int32 [slices, N] = index of its first block
    A block = up to 64 consecutive elements of one (slice, row, cell)."""
    assert indices.dtype == torch.int32 and indices.dim() == 3 and indices.shape[0] == 1
    assert slices in (8, 16)
    S, SLICE_BITS = slices, INDEX_BITS - (3 if slices == 8 else 4)
    dev = indices.device
    N, G = indices.shape[1], group_size
    nbytes = 3 if residual else 2
    by = indices[0].contiguous().view(torch.uint8).reshape(N, -1)[:, :nbytes * G].reshape(N, G, nbytes).to(torch.int64)
    idx = by[:, :, 0] | (by[:, :, 1] << 8)                                    # [N, G] main index per column
    ridx = by[:, :, 2] if residual else torch.zeros_like(idx)
    sl = idx >> SLICE_BITS
    col = torch.arange(G, device=dev, dtype=torch.int64)
    cell = col >> CELL_BITS
    C = int(cell[-1].item()) + 1
    seg = sl * C + cell[None, :]                                              # (slice, cell) of every element
    order = torch.argsort(seg * G + col[None, :], dim=1)                      # a row's columns by (slice, cell, column)
    seg_sorted = torch.gather(seg, 1, order)
    word = (order & ((1 << CELL_BITS) - 1)) | ((torch.gather(idx, 1, order) & ((1 << SLICE_BITS) - 1)) << CELL_BITS) | \
        (torch.gather(ridx, 1, order) << 24)
    cnt = torch.zeros(N, S * C, dtype=torch.int64, device=dev)                # elements per (row, slice, cell)
    cnt.scatter_add_(1, seg, torch.ones_like(seg))
    seg_row_start = torch.cumsum(cnt, 1) - cnt                                # first position of (slice, cell) in the sorted row
    cnt_sn = cnt.reshape(N, S, C).sum(2).t().contiguous()                     # [S, N] elements of (slice, row)
    goff = (torch.cumsum(cnt_sn.reshape(-1), 0) - cnt_sn.reshape(-1)).reshape(S, N)   # first element of (slice, row)
    slice_row_start = seg_row_start.reshape(N, S, C)[:, :, 0]                 # [N, S] first position of slice s in the sorted row
    rows = torch.arange(N, device=dev)[:, None].expand(N, G)
    s_sorted = seg_sorted // C
    dest = goff[s_sorted, rows] + (col[None, :] - slice_row_start[rows, s_sorted])
    total_e = N * G
    elems = torch.zeros(total_e + 64, dtype=torch.int64, device=dev)          # (+ 64: a block load never leaves the array)
    elems[dest.reshape(-1)] = word.reshape(-1)
    # blocks: segments flattened in (slice, row, cell) order
    cnt_f = cnt.reshape(N, S, C).permute(1, 0, 2).reshape(-1)                 # [S * N * C]
    seg_e0 = (goff[:, :, None] + (seg_row_start.reshape(N, S, C) - slice_row_start[:, :, None]).permute(1, 0, 2)).reshape(-1)
    nb_f = (cnt_f + 63) // 64
    nsegs = nb_f.numel()
    segmap = torch.repeat_interleave(torch.arange(nsegs, device=dev), nb_f)
    j = torch.arange(segmap.numel(), device=dev) - (torch.cumsum(nb_f, 0) - nb_f)[segmap]
    bstart = seg_e0[segmap] + 64 * j
    valid = torch.clamp(cnt_f[segmap] - 64 * j, max=64)
    bmeta = valid | ((segmap % C) << 8)
    blocks_sn = nb_f.reshape(S, N, C).sum(2)
    first_sn = (torch.cumsum(blocks_sn.reshape(-1), 0) - blocks_sn.reshape(-1)).reshape(S, N)
    if bstart.numel() == 0:
        bstart, bmeta = torch.zeros(1, dtype=torch.int64, device=dev), torch.zeros(1, dtype=torch.int64, device=dev)
    elems32 = torch.where(elems >= (1 << 31), elems - (1 << 32), elems).to(torch.int32)
    return (elems32, bstart.to(torch.int32).contiguous(), bmeta.to(torch.int32).contiguous(),
            blocks_sn.to(torch.int32).contiguous(), first_sn.to(torch.int32).contiguous())


def rows_per_wave_for(n_rows: int, slices: int = 8, workgroups: int = 256) -> int:
    """consecutive rows per wave so that slices x row blocks of 16 waves give about `workgroups` workgroups"""
    r = max(1, (n_rows * slices + 16 * workgroups - 1) // (16 * workgroups))
    return min(r, 64)


class SlicedGemv:
    """One-token forward of a v8-k65536-0 `VQuantLinear` over its sliced layout."""

    def __init__(self, layer, rows_per_wave: int = 0):
        self.layer = layer
        cache = layer._descriptor()
        self.desc, self.dev = cache[1], cache[3]
        self.slices = B.lib().vptq_sliced_layout_supported(self.desc)
        if not self.slices:
            raise ValueError("the sliced layout serves v8-k65536-0 / v8-k65536-256 layers without a permutation, "
                             "group_size <= 32768")
        residual = bool(layer.enable_residual)
        self.elems, self.bstart, self.bmeta, self.blocks, self.first = build_sliced_layout(
            layer.indices.data, layer.group_size, self.slices, residual)
        self.layout = B.SlicedLayout(self.elems.data_ptr(), self.bstart.data_ptr(), self.bmeta.data_ptr(),
                                     self.blocks.data_ptr(), self.first.data_ptr(),
                                     rows_per_wave or rows_per_wave_for(self.blocks.shape[1], self.slices), self.slices)
        nb = B.lib().vptq_quant_gemv_sliced_workspace_bytes(self.desc)
        self.ws = torch.zeros(nb, dtype=torch.uint8, device=self.dev)   # (arrival counters: zero once, every call leaves them zero)
        self._fn = B.lib().vptq_quant_gemv_sliced
        self._lay_ref = C.byref(self.layout)
        self._ws_ptr, self._ws_bytes = self.ws.data_ptr(), self.ws.numel()
        self._dtype = cache[7]
        self._dev_index = cache[8]
        self.extra_bytes = (self.elems.numel() + self.bstart.numel() * 2 + self.blocks.numel() * 2) * 4

    def __call__(self, x: torch.Tensor, out: torch.Tensor = None, flags: int = 0) -> torch.Tensor:
        lay = self.layer
        # (the checks of VQuantLinear._check_activation, against cached values: this is the per-token path)
        if x.shape[-1] != lay.in_features or x.numel() != lay.in_features:
            raise ValueError("the sliced path takes one token of in_features values")
        if x.dtype != self._dtype or x.device != self.dev:
            x = lay._check_activation(x)
        if not x.is_contiguous():
            x = x.contiguous()
        if out is None:
            out = torch.empty(x.shape[:-1] + (lay.out_features,), dtype=torch.float32 if (flags & B.GEMV_OUT_F32) else self._dtype,
                              device=self.dev)
        if torch.cuda.current_device() != self._dev_index:
            with torch.cuda.device(self.dev):
                rc = self._fn(self.desc, self._lay_ref, x.data_ptr(), out.data_ptr(), flags, self._ws_ptr, self._ws_bytes,
                              B.current_stream_ptr(self.dev))
        else:
            rc = self._fn(self.desc, self._lay_ref, x.data_ptr(), out.data_ptr(), flags, self._ws_ptr, self._ws_bytes,
                          B.current_stream_ptr(self.dev))
        if rc:
            B.check(rc, "vptq_quant_gemv_sliced")
        return out
