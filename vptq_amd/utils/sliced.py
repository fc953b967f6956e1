"""Load-time derived layout for large-codebook layers (v = 8, k = 65536, residual none, 256 or 65536: "v8-k65536-0",
"v8-k65536-256", "v8-k65536-65536"), the formats of most published VPTQ checkpoints, and the one-token GEMV over it
(`vptq_quant_gemv_sliced`, vptq_amd/csrc/gemv_sliced.hip).

The reference gathers centroid rows from the 1 MiB codebook through the caches for every index
(csrc/kernels/quant_gemv.cuh:11-186); on MI355X that is bound by the L2 -> L1 fill rate (39.6 us per 8192^2
layer).  Bucketing every row's elements ONCE by the top 3 (wide layers: 4) bits of their index
lets a workgroup keep its 8192- (4096-) entry slice of the codebook in LDS.  The state-dict tensors are untouched (they stay
the contract, and the many-token / dequant paths keep using them); the derived tensors cost 2x (T = 16) / 1.7x
(T = 24: 5 instead of 3 bytes per element) the packed indices in device memory on top.

    sl = SlicedGemv(layer)          # builds the layout (torch, on the layer's device)
    y = sl(x)                       # one token; same result as layer(x) within the parity bar

Round 5: `SlicedGemv(layer, exact=True)` - the reference's roundings per weight (the product default) over ONE layout bucketed by
the main index (a residual codebook other than v8's 256-entry one rides along as 16-bit indices, its entries gathered from L2);
layers too wide for that arithmetic's 6 bytes of LDS per column in one piece (28672 columns) as 2 - 3 COLUMN PARTS with a layout
each (`exact_column_parts`, `part_desc`); `forward_tokens` for 2 - 8 tokens: 2 / 3 in one pass of the one-token kernel where the
operands fit - whole (`tokens_window_parts` = 1) or in window parts (2 / 4) -, else column phases on the matrix pipe.
"""
from __future__ import annotations

import ctypes as C

import torch

from vptq_amd import _backend as B

INDEX_BITS = 16   # 65536 main centroids
WINDOWS = 4       # column windows the lists are ordered by (VPTQ_SLICED_WINDOWS in include/vptq_hip.h)


def window_cols(G: int) -> int:
    """columns per window: G / WINDOWS rounded up to a multiple of 8 (the last window takes what is left)"""
    return (G + WINDOWS * 8 - 1) // (WINDOWS * 8) * 8


def split_index_streams(indices: torch.Tensor, group_size: int, res_bits: int, index_bits: int = INDEX_BITS):
    """the layer's packed int32 `indices` [1, N, row_words] -> (main index [N, G] int64, residual index [N, G] int64 or None):
    a little-endian bit stream per row, element g at bits [T g, T g + T) with value (residual index << index_bits) | main
    index, T = index_bits + res_bits (vptq/utils/pack.py:26-89)."""
    assert indices.dtype == torch.int32 and indices.dim() == 3 and indices.shape[0] == 1
    from vptq_amd.utils.pack import unpack_index_tensor
    idx, ridx = unpack_index_tensor(indices, index_bits, group_size, res_bits, group_size, res_mask_bits=res_bits)
    return idx[0], (None if ridx is None else ridx[0])


def build_sliced_layout(indices: torch.Tensor, group_size: int, slices: int = 8, residual: bool = False):
    """indices: the layer's packed int32 `indices` [1, N, row_words] (T = 16 without a residual codebook, 24 with 256
    residual centroids).  slices: 8 or 16 (vptq_sliced_layout_supported tells).
    Returns (elems uint32-as-int32 [blocks * 64], blocks int32 [slices, N], first int32 [slices, N], res uint8
    [like elems] or None, wstart int32 [slices, N, WINDOWS + 1]) as described in include/vptq_hip.h (VptqSlicedLayout)."""
    idx, ridx = split_index_streams(indices, group_size, 8 if residual else 0)
    return layout_from_indices(idx, slices, ridx)


def layout_from_indices(idx: torch.Tensor, slices: int = 8, ridx: torch.Tensor = None, index_bits: int = INDEX_BITS,
                        whole_table: bool = False, side_dtype=torch.uint8):
    """idx [N, G] int64 (values < 2^index_bits): the index each element gathers with from the table this layout is for;
    ridx: a side index carried along per element, or None - 8 bits (the 256-entry residual table of v = 8, held in LDS) or, for
    the reference's roundings with any other residual codebook, 16 bits as int16 bit patterns (`side_dtype=torch.int16`: the
    residual entry is gathered from device memory).  Elements are
    bucketed by the top log2(slices) bits of idx and the element word carries idx inside its slice; or - whole_table:
    every workgroup of the table holds the whole (small) table - split into equal column ranges, the word carrying idx itself."""
    assert slices in (8, 16, 32) and 1 <= index_bits <= 16
    SLICES = slices
    lg = {8: 3, 16: 4, 32: 5}[slices]
    SLICE_BITS = max(index_bits - lg, 0)
    dev = idx.device
    N, G = idx.shape
    residual = ridx is not None
    col = torch.arange(G, device=dev, dtype=torch.int64)
    if whole_table:
        # every workgroup of the table holds all of it, so ANY split of a row's elements is valid: equal column ranges
        # (bucketing a 4- or 64-entry table by its index would leave most of the slices - workgroups - empty)
        sl = (col * SLICES // G)[None, :].expand(N, G).contiguous()
    else:
        sl = idx >> SLICE_BITS
    # Order inside a (row, slice) list is free (a sum): arrange it so that 16 CONSECUTIVE elements - the lanes one
    # pass of the kernel's ds_read_b128 gather serves - hit 16 different LDS bank groups (entry & 15): elements are
    # ranked inside their (slice, entry & 15) class and laid out rank-major, i.e. one element of every class in turn.
    # (Column order gave 3-way conflicts on average: SQ_LDS_BANK_CONFLICT = 60 % of the LDS cycles.)
    local = idx if whole_table else idx & ((1 << SLICE_BITS) - 1)
    cls = local & 15
    # ... and, before that, by COLUMN WINDOW (WINDOWS equal ranges of window_cols(G) columns): the kernel for 2 - 4 tokens
    # (gemv_sliced_tok.hip) stages one window of the activations at a time - T tokens of G / T columns fill the room one token of
    # G columns has - and walks the window's part of every list; `wstart` tells where it begins.  One token ignores it.
    wcols = window_cols(G)
    win = torch.clamp(col // wcols, max=WINDOWS - 1)[None, :].expand(N, G)
    sw = sl * WINDOWS + win
    order1 = torch.argsort((sw * 16 + cls) * G + col[None, :], dim=1)
    seg1 = torch.gather(sw * 16 + cls, 1, order1)                                       # sorted (slice, window, class) id
    cnt1 = torch.zeros(N, SLICES * WINDOWS * 16, dtype=torch.int64, device=dev)
    cnt1.scatter_add_(1, seg1, torch.ones_like(seg1))
    rank1 = col[None, :] - torch.gather(torch.cumsum(cnt1, 1) - cnt1, 1, seg1)          # rank inside the class
    # rank-major while every class still has an element (rows of 16 different classes); what is left over - the surplus of
    # the classes that have more than the smallest one - is spread evenly over the rest of the (slice, window) list (an
    # element's place = (its rank + 1/2) / its class's surplus of the way through), not bunched class by class at its end
    cnt3 = cnt1.reshape(N, SLICES * WINDOWS, 16)
    full = cnt3.min(2).values                                                               # complete rows per (slice, window)
    rest = cnt3.sum(2) - 16 * full
    b1 = seg1 >> 4
    full_e, rest_e, cnt_e = torch.gather(full, 1, b1), torch.gather(rest, 1, b1), torch.gather(cnt1, 1, seg1)
    spread = 16 * full_e + ((rank1 - full_e).double() + 0.5) * rest_e.double() / (cnt_e - full_e).clamp(min=1).double()
    place = torch.where(rank1 < full_e, (rank1 * 16 + (seg1 & 15)).double(), spread)
    order2 = torch.argsort((b1 * (2 * G)).double() + place + (seg1 & 15).double() / 64.0, dim=1)   # (slice, window, place)
    order = torch.gather(order1, 1, order2)
    s_sorted = torch.gather(sl, 1, order)
    word = order | (torch.gather(local, 1, order) << 16)                                # column | local << 16
    counts = torch.zeros(N, SLICES, dtype=torch.int64, device=dev)
    counts.scatter_add_(1, sl, torch.ones_like(sl))
    seg_start = torch.cumsum(counts, 1) - counts                                        # first position of (n, s) in the sorted row
    bs = 64                                                                             # elements per block
    blocks = (counts + bs - 1) // bs                                                    # [N, 8]
    blocks_sn = blocks.t().contiguous()                                                 # [8, N]
    first_sn = (torch.cumsum(blocks_sn.reshape(-1), 0) - blocks_sn.reshape(-1)).reshape(SLICES, N)
    total = int(blocks_sn.sum().item())
    elems = torch.full((max(total, 1) * bs,), G, dtype=torch.int64, device=dev)         # padding: column G, local 0
    pos = col[None, :] - torch.gather(seg_start, 1, s_sorted)                           # rank inside its (n, s) list
    rows = torch.arange(N, device=dev)[:, None].expand(N, G)
    dest = first_sn[s_sorted, rows] * bs + pos
    elems[dest.reshape(-1)] = word.reshape(-1)
    elems32 = (elems & 0xffffffff).to(torch.int64)
    elems32 = torch.where(elems32 >= (1 << 31), elems32 - (1 << 32), elems32).to(torch.int32)
    res = None
    if residual:
        res = torch.zeros(elems.numel(), dtype=side_dtype, device=dev)
        rv = torch.gather(ridx, 1, order).reshape(-1)
        if side_dtype == torch.int16:       # (uint16 bit patterns)
            rv = torch.where(rv >= 32768, rv - 65536, rv)
        res[dest.reshape(-1)] = rv.to(side_dtype)
    # wstart [slices][N][WINDOWS + 1]: position inside the (s, n) list at which window w begins; [WINDOWS] = the list's length
    cntw = torch.zeros(N, SLICES * WINDOWS, dtype=torch.int64, device=dev)
    cntw.scatter_add_(1, sw, torch.ones_like(sw))
    cntw = cntw.reshape(N, SLICES, WINDOWS)
    wstart = torch.zeros(N, SLICES, WINDOWS + 1, dtype=torch.int64, device=dev)
    wstart[:, :, 1:] = torch.cumsum(cntw, 2)
    wstart = wstart.permute(1, 0, 2).to(torch.int32).contiguous()
    return elems32, blocks_sn.to(torch.int32).contiguous(), first_sn.to(torch.int32).contiguous(), res, wstart


def rows_per_wave_for(n_rows: int, slices: int = 8, workgroups: int = 256) -> int:
    """consecutive rows per wave so that slices x row blocks of 16 waves give about `workgroups` workgroups"""
    r = max(1, (n_rows * slices + 16 * workgroups - 1) // (16 * workgroups))
    return min(r, 64)


def part_desc(desc, c0: int, c1: int):
    """copy of a layer's descriptor that stands for its columns [c0, c1) alone (`VPTQ_GEMV_COLUMN_PARTS`, include/vptq_hip.h): widths
    and the column-order tensors advanced to c0 (16-bit elements).  The tensors stay owned by the layer's own descriptor."""
    d = B.LayerDesc.from_buffer_copy(desc)
    d.in_features = d.group_size = c1 - c0
    for f in ("weight_scale", "weight_bias", "perm", "scale_permuted", "bias_permuted"):
        p = getattr(desc, f)
        if p:
            setattr(d, f, p + 2 * c0)
    return d


def exact_column_parts(desc, group_size: int):
    """(parts, slices) with which the reference's roundings are served over sliced layouts: (1, n) where the layer fits in one
    piece, (2 or 3, n) where equal column parts of a multiple of 8 columns do (28672-column layers: 2 x 14336), (0, 0) else"""
    import os
    least = int(B.tune_env("VPTQ_SLICED_PARTS", "1") or 1)     # (A/B: at least this many parts where the columns divide)
    n = B.lib().vptq_sliced_layout_supported_for(desc, B.GEMV_EXACT)
    if n and least <= 1:
        return 1, n
    for parts in (2, 3):
        if parts < least:
            continue
        if group_size % (8 * parts) == 0:
            n = B.lib().vptq_sliced_layout_supported_for(part_desc(desc, 0, group_size // parts), B.GEMV_EXACT)
            if n and n * parts <= 127:
                return parts, n
    if least > 1:
        n = B.lib().vptq_sliced_layout_supported_for(desc, B.GEMV_EXACT)
        return (1, n) if n else (0, 0)
    return 0, 0


class SlicedGemv:
    """One-token forward of a v8-k65536-0 `VQuantLinear` over its sliced layout."""

    def __init__(self, layer, rows_per_wave: int = 0, exact: bool = False, selective: bool = False):
        """exact: the reference's roundings per weight (`VPTQ_GEMV_EXACT`, the product default arithmetic) instead of the
        folded form - layers without a residual codebook or with the 256-entry one of v = 8; the layout then has the slice
        count that arithmetic needs (scale, bias and x of every column sit beside the slice in LDS).
        selective (round 6, one token): the FOLDED layouts with `VPTQ_GEMV_SELECTIVE` - a pre-pass finds the blocks of 128 columns an
        activation dominates, the folded launch reads them as zeros and adds their exact products (gemv_hot.hip)."""
        self.layer = layer
        self.exact = bool(exact)
        self.selective = bool(selective) and not self.exact
        self._flags = B.GEMV_EXACT if self.exact else (B.GEMV_SELECTIVE if self.selective else 0)
        if self.selective and not B.lib().vptq_quant_gemv_sliced_selective_supported(layer._descriptor()[1]):
            raise ValueError("selective roundings over the sliced layouts: fp16 layers with scale and bias")
        cache = layer._descriptor()
        self.desc, self.dev = cache[1], cache[3]
        self.parts = 1
        if self.exact:   # (a layer too wide for 6 bytes of LDS per column in one piece: equal column parts, one layout each)
            self.parts, self.slices = exact_column_parts(self.desc, layer.group_size)
        else:
            self.slices = B.lib().vptq_sliced_layout_supported_for(self.desc, 0)
        if not self.slices:
            raise ValueError("the sliced layout serves v = 8 / 16 layers with 16384 ... 65536 main centroids, group_size <= 32768"
                             " (reference roundings: one table, up to ~16000 columns)")
        kr = layer.num_res_centroids if layer.enable_residual else 0
        ib = int(layer.num_centroids).bit_length() - 1
        rb = int(kr).bit_length() - 1 if kr else 0
        # (the reference's roundings need c and r in one lane: always ONE layout, bucketed by the main index; a residual codebook
        # other than v8's 256-entry one rides along as a 16-bit side stream and its entries are gathered from device memory)
        n_tables = 1 if self.exact else B.lib().vptq_sliced_layout_tables(self.desc)
        side16 = self.exact and kr > 0 and not (layer.vector_len == 8 and kr == 256)
        self._side16 = side16
        # a second table whose slice would be under 16 KiB is held WHOLE by each of its workgroups while it fits (the
        # library decides: the kernel's LDS budget)
        whole = [False, n_tables == 2 and bool(B.lib().vptq_sliced_layout_whole_table(self.desc, 1))]
        idx, ridx = split_index_streams(layer.indices.data, layer.group_size, rb, ib)
        if self.parts > 1:
            w = layer.group_size // self.parts
            self._tensors = [layout_from_indices(idx[:, p * w:(p + 1) * w].contiguous(), self.slices,
                                                 ridx[:, p * w:(p + 1) * w].contiguous() if kr else None, ib,
                                                 side_dtype=torch.int16 if side16 else torch.uint8) for p in range(self.parts)]
            self._part_descs = (B.LayerDesc * self.parts)(*[part_desc(self.desc, p * w, (p + 1) * w) for p in range(self.parts)])
            whole = [False] * self.parts
        elif n_tables == 2:
            # (c + r) s x = c s x + r s x: the residual codebook is a second table with a layout bucketed by ITS index
            self._tensors = [layout_from_indices(idx, self.slices, None, ib), layout_from_indices(ridx, self.slices, None, rb, whole[1])]
        else:
            self._tensors = [layout_from_indices(idx, self.slices, ridx if kr else None, ib,
                                                 side_dtype=torch.int16 if side16 else torch.uint8)]
        del idx, ridx
        self._whole = whole[:len(self._tensors)]
        self.elems, self.blocks, self.first, self.res, self.wstart = self._tensors[0]
        # (a two-table layer runs 2 x slices workgroups per row block in its one launch)
        rpw = rows_per_wave or rows_per_wave_for(self.blocks.shape[1], self.slices * len(self._tensors))
        self.layout = (B.SlicedLayout * len(self._tensors))(*[
            B.SlicedLayout(e.data_ptr(), b.data_ptr(), f.data_ptr(), r.data_ptr() if r is not None else None, rpw, 1, self.slices, int(w),
                           ws.data_ptr())
            for (e, b, f, r, ws), w in zip(self._tensors, self._whole)])
        self._ws_bytes = B.lib().vptq_quant_gemv_sliced_workspace_bytes(self._part_descs[0] if self.parts > 1 else self.desc)
        if self.selective:   # (+ header, x with the hot blocks zeroed, the hot blocks' exact products)
            self._ws_bytes = B.lib().vptq_quant_gemv_sliced_workspace_bytes_for(self.desc, self._flags)
        if self.parts > 1:
            self._pp = ((C.c_void_p * self.parts)(), (C.c_void_p * self.parts)(), (C.c_size_t * self.parts)(*([self._ws_bytes] * self.parts)))
        # partial sums + arrival counters, ONE PER STREAM (two streams - or a graph replay next to an eager call on
        # another stream - running the same layer would race on a shared one); zeroed once, every call leaves the
        # counters zero
        self._ws = {}
        self._ws_tok = {}   # stream -> workspace of the 2 - 8 token kernel
        self._retired = []  # workspaces that were ever handed to a launch stay alive as long as the layer does
        self._fn = B.lib().vptq_quant_gemv_sliced
        self._fn_tok = B.lib().vptq_quant_gemv_sliced_tokens
        self._lay_ref = self.layout   # (an array of 1 or 2 structs: passed as a pointer to the first)
        self._dtype = cache[7]
        self._dev_index = cache[8]
        self.extra_bytes = sum(e.numel() * (4 + (r.element_size() if r is not None else 0)) + b.numel() * 8 for e, b, f, r, _ in self._tensors)

    def _workspace(self, stream_ptr: int):
        ws = self._ws.get(stream_ptr)
        if ws is None:
            if torch.cuda.is_current_stream_capturing():
                return None   # (no allocation + memset inside a capture: warm the layer up on the capture stream first)
            ws = torch.zeros(self._ws_bytes, dtype=torch.uint8, device=self.dev)
            self._ws[stream_ptr] = ws
        return ws

    def __call__(self, x: torch.Tensor, out: torch.Tensor = None, flags: int = 0):
        """y, or None where this call cannot take the sliced kernel (activation not 16-byte aligned, no workspace for a
        capturing stream, the library says "unsupported"): the caller then takes the regular route."""
        lay = self.layer
        # (the checks of VQuantLinear._check_activation, against cached values: this is the per-token path)
        if x.shape[-1] != lay.in_features or x.numel() != lay.in_features:
            raise ValueError("the sliced path takes one token of in_features values")
        if x.dtype != self._dtype or x.device != self.dev:
            x = lay._check_activation(x)
        if not x.is_contiguous():
            x = x.contiguous()
        if x.data_ptr() & 15:
            return None
        if torch.cuda.current_device() != self._dev_index:
            with torch.cuda.device(self.dev):
                return self._launch(x, out, flags)
        return self._launch(x, out, flags)

    def tokens_supported(self, tokens: int) -> bool:
        """does the library's kernel for 2 - 4 tokens over these layouts take this layer (its activations must fit the LDS
        beside the slice in at most 4 column phases)?"""
        if self.exact and self.parts > 1:    # (column parts: 2 / 3 tokens where every part takes them in one pass, nothing else)
            return self.tokens_one_pass(tokens)
        if self.exact and self._side16:      # (the reference's roundings over two-table formats: 2 / 3 tokens in one pass, nothing else)
            return self.tokens_one_pass(tokens)
        if self.selective:                   # (one token: several tokens of such a layer take the module's regular route)
            return False
        return bool(B.lib().vptq_quant_gemv_sliced_tokens_supported_for(self.desc, self._lay_ref, int(tokens), self._flags))

    def tokens_one_pass(self, tokens: int) -> bool:
        """(reference roundings) does the library take these 2 / 3 tokens in ONE PASS of the one-token kernel (gemv_sliced.hip, TOK:
        slice + (2 tokens + 4) bytes per column fit the LDS) - the route that needs no column windows?"""
        if not self.exact or not 2 <= tokens <= 3:
            return False
        return self.tokens_window_parts(tokens) >= 1

    def tokens_window_parts(self, tokens: int) -> int:
        """0, or in how many window parts the library takes these tokens in one pass: 1 = every column staged beside the slice; 2 / 4 =
        that many workgroups per (slice, row block), each with its column windows (`vptq_quant_gemv_sliced_tokens_one_pass`)"""
        if not self.exact or not 2 <= tokens <= 3:
            return 0
        key = ("_one_pass", tokens)
        n = self.__dict__.get(key)
        if n is None:
            d = self._part_descs[0] if self.parts > 1 else self.desc     # (column parts: every part is asked - they have one width)
            n = int(B.lib().vptq_quant_gemv_sliced_tokens_one_pass(d, int(tokens), self._flags))
            if n and not B.lib().vptq_quant_gemv_sliced_tokens_supported_for(d, self._lay_ref, int(tokens), self._flags):
                n = 0
            self.__dict__[key] = n
        return n

    def forward_tokens(self, x: torch.Tensor, out: torch.Tensor = None, flags: int = 0):
        """2 - 4 tokens in ONE launch (`vptq_quant_gemv_sliced_tokens`, gemv_sliced_tok.hip): x [..., in_features] with 2 - 4
        rows, contiguous.  Returns y, or None where the call cannot be served (the caller takes the regular route)."""
        lay = self.layer
        tokens = x.numel() // lay.in_features
        if x.shape[-1] != lay.in_features or not 2 <= tokens <= 8:
            raise ValueError("forward_tokens takes 2 - 8 tokens of in_features values")
        if self.selective or (self.exact and (self.parts > 1 or self._side16) and not self.tokens_one_pass(tokens)):
            return None
        if x.dtype != self._dtype or x.device != self.dev:
            x = lay._check_activation(x)
        if not x.is_contiguous():
            x = x.contiguous()
        if x.data_ptr() & 15 or (lay.in_features * x.element_size()) & 15:
            return None
        if torch.cuda.current_device() != self._dev_index:
            with torch.cuda.device(self.dev):
                return self._launch_tokens(x, out, flags, tokens)
        return self._launch_tokens(x, out, flags, tokens)

    def _tokens_workspace(self, stream_ptr: int, tokens: int = 4):
        """the workspace of the token kernel for this stream, or None inside a capture on a stream the layer has not run on.
        Sized ONCE for the most tokens the library serves for this layer (8, else 4 - a workspace for T tokens serves fewer)
        and never replaced or freed afterwards: a hipGraph captured on this stream holds its address (one workspace per
        stream = per capture; a graph replayed on another stream than the one it was captured on races with eager calls on
        the first - replay where you captured)."""
        ws = self._ws_tok.get(stream_ptr)
        need = 8 if tokens > 4 else 4
        if ws is None or ws[1] < need:
            if torch.cuda.is_current_stream_capturing():
                return None
            wd = self._part_descs[0] if self.parts > 1 else self.desc
            size, nbytes = 8, B.lib().vptq_quant_gemv_sliced_tokens_workspace_bytes(wd, 8)
            if not nbytes:
                size, nbytes = 4, B.lib().vptq_quant_gemv_sliced_tokens_workspace_bytes(wd, 4)
            if not nbytes or size < need:
                return None
            if ws is not None:
                self._retired.append(ws[0])   # (cannot happen with the sizing above; kept alive if it ever does)
            ws = (torch.zeros(nbytes, dtype=torch.uint8, device=self.dev), size)
            self._ws_tok[stream_ptr] = ws
        return ws[0]

    def _launch_tokens(self, x, out, flags, tokens):
        lay = self.layer
        sp = B.current_stream_ptr(self.dev)
        ws = self._tokens_workspace(sp, tokens)
        if ws is None:
            return None
        if out is None:
            out = torch.empty(x.shape[:-1] + (lay.out_features,),
                              dtype=torch.float32 if (flags & B.GEMV_OUT_F32) else self._dtype, device=self.dev)
        if self.parts > 1:   # column parts of one layer: one grouped launch, shared output and workspace
            yp, wp, _ = self._pp
            for i in range(self.parts):
                yp[i] = out.data_ptr()
                wp[i] = ws.data_ptr()
            wb = (C.c_size_t * self.parts)(*([ws.numel()] * self.parts))
            rc = B.lib().vptq_quant_gemv_sliced_tokens_grouped(self._part_descs, self._lay_ref, self.parts, x.data_ptr(), yp, tokens,
                                                               flags | self._flags | B.GEMV_COLUMN_PARTS, wp, wb, sp)
        else:
            rc = self._fn_tok(self.desc, self._lay_ref, x.data_ptr(), out.data_ptr(), tokens, flags | self._flags, ws.data_ptr(), ws.numel(), sp)
        if rc == B.E_UNSUPPORTED:
            return None
        if rc:
            ws.zero_()   # (a launch that did not happen or did not finish may have left arrival counters behind; the buffer
            B.check(rc, "vptq_quant_gemv_sliced_tokens")   # itself stays: a captured graph may hold its address)
        return out

    def _launch(self, x, out, flags):
        sp = B.current_stream_ptr(self.dev)
        ws = self._workspace(sp)
        if ws is None:
            return None
        if out is None:
            out = torch.empty(x.shape[:-1] + (self.layer.out_features,),
                              dtype=torch.float32 if (flags & B.GEMV_OUT_F32) else self._dtype, device=self.dev)
        if self.parts > 1:   # column parts of one layer: one grouped launch, shared output and accumulator words
            yp, wp, wb = self._pp
            for i in range(self.parts):
                yp[i] = out.data_ptr()
                wp[i] = ws.data_ptr()
            rc = B.lib().vptq_quant_gemv_sliced_grouped(self._part_descs, self._lay_ref, self.parts, x.data_ptr(), yp,
                                                        flags | self._flags | B.GEMV_COLUMN_PARTS, wp, wb, sp)
        else:
            rc = self._fn(self.desc, self._lay_ref, x.data_ptr(), out.data_ptr(), flags | self._flags, ws.data_ptr(), self._ws_bytes, sp)
        if rc == B.E_UNSUPPORTED:
            return None
        if rc:
            # (a launch that did not happen or did not finish may have left accumulator words behind; the buffer itself
            # stays: a captured graph may hold its address)
            ws.zero_()
            B.check(rc, "vptq_quant_gemv_sliced")
        return out


class SlicedGroupGemv:
    """One token through up to 3 sibling layers (q / k / v, gate / up: one format, one input width, the SAME activation)
    in ONE launch of the sliced kernel (`vptq_quant_gemv_sliced_grouped`): the fixed part of a sliced launch - boundary,
    slice copy, staging, the cross-slice hand-over: ~7 of the 10 us of a 4096 x 4096 layer - is paid once.  `members`:
    the layers' `SlicedGemv` objects (their tensors are shared, only the rows per wave are the group's)."""

    def __init__(self, members):
        self.members = list(members)
        n = len(self.members)
        m0 = self.members[0]
        tables = len(m0.layout)
        kind = lambda m: (m.layer.vector_len, m.layer.num_centroids,   # noqa: E731
                          m.layer.num_res_centroids if m.layer.enable_residual else 0, tuple(m._whole), m.exact)
        if not 1 <= n <= 3 or any(m.parts > 1 or getattr(m, "selective", False) for m in self.members) or any(len(m.layout) != tables or m.slices != m0.slices or m._dtype != m0._dtype or m.dev != m0.dev or
                                  m.layer.in_features != m0.layer.in_features or kind(m) != kind(m0) for m in self.members):
            raise ValueError("a sliced group takes 1..3 layers of one format (vector length, codebook sizes), dtype, device and input width")
        rpw = rows_per_wave_for(sum(m.blocks.shape[1] for m in self.members), m0.slices * tables)   # one round of workgroups over ALL layers
        structs = [B.SlicedLayout(e.data_ptr(), b.data_ptr(), f.data_ptr(), r.data_ptr() if r is not None else None, rpw, 1, m.slices, int(w),
                                  ws.data_ptr())
                   for m in self.members for (e, b, f, r, ws), w in zip(m._tensors, m._whole)]
        self.layouts = (B.SlicedLayout * len(structs))(*structs)
        self.descs = (B.LayerDesc * n)(*[m.desc for m in self.members])
        self._yp, self._wp = (C.c_void_p * n)(), (C.c_void_p * n)()
        self._wb = (C.c_size_t * n)(*[m._ws_bytes for m in self.members])
        self._fn = B.lib().vptq_quant_gemv_sliced_grouped
        self.dev, self._dtype, self._dev_index = m0.dev, m0._dtype, m0._dev_index
        self.exact, self._flags, self._side16 = m0.exact, m0._flags, m0._side16

    def __call__(self, x: torch.Tensor):
        """list of outputs (one per member), or None where the call cannot take the sliced kernel (as SlicedGemv.__call__)"""
        lay = self.members[0].layer
        if x.shape[-1] != lay.in_features or x.numel() != lay.in_features:
            raise ValueError("the sliced path takes one token of in_features values")
        if x.dtype != self._dtype or x.device != self.dev:
            x = lay._check_activation(x)
        if not x.is_contiguous():
            x = x.contiguous()
        if x.data_ptr() & 15:
            return None
        if torch.cuda.current_device() != self._dev_index:
            with torch.cuda.device(self.dev):
                return self._launch(x)
        return self._launch(x)

    def tokens_supported(self, tokens: int) -> bool:
        return all(m.tokens_supported(tokens) for m in self.members)

    def forward_tokens(self, x: torch.Tensor):
        """2 - 4 tokens through every member in ONE launch (`vptq_quant_gemv_sliced_tokens_grouped`): list of outputs, or None
        where the call cannot be served (as SlicedGemv.forward_tokens)"""
        lay = self.members[0].layer
        tokens = x.numel() // lay.in_features
        if x.shape[-1] != lay.in_features or not 2 <= tokens <= 8:
            raise ValueError("forward_tokens takes 2 - 8 tokens of in_features values")
        if self.exact and self._side16 and not all(m.tokens_one_pass(tokens) for m in self.members):
            return None
        if x.dtype != self._dtype or x.device != self.dev:
            x = lay._check_activation(x)
        if not x.is_contiguous():
            x = x.contiguous()
        if x.data_ptr() & 15 or (lay.in_features * x.element_size()) & 15:
            return None
        if torch.cuda.current_device() != self._dev_index:
            with torch.cuda.device(self.dev):
                return self._launch_tokens(x, tokens)
        return self._launch_tokens(x, tokens)

    def _launch_tokens(self, x, tokens):
        sp = B.current_stream_ptr(self.dev)
        n = len(self.members)
        wss = []
        for m in self.members:
            ws = m._tokens_workspace(sp, tokens)
            if ws is None:
                return None
            wss.append(ws)
        ys = [torch.empty(x.shape[:-1] + (m.layer.out_features,), dtype=self._dtype, device=self.dev) for m in self.members]
        wb = (C.c_size_t * n)(*[w.numel() for w in wss])
        for i, (y, w) in enumerate(zip(ys, wss)):
            self._yp[i] = y.data_ptr()
            self._wp[i] = w.data_ptr()
        rc = B.lib().vptq_quant_gemv_sliced_tokens_grouped(self.descs, self.layouts, n, x.data_ptr(), self._yp, tokens, self._flags, self._wp, wb, sp)
        if rc == B.E_UNSUPPORTED:
            return None
        if rc:
            for w in wss:
                w.zero_()
            B.check(rc, "vptq_quant_gemv_sliced_tokens_grouped")
        return ys

    def _launch(self, x):
        sp = B.current_stream_ptr(self.dev)
        wss = [m._workspace(sp) for m in self.members]
        if any(w is None for w in wss):
            return None
        ys = [torch.empty(x.shape[:-1] + (m.layer.out_features,), dtype=self._dtype, device=self.dev) for m in self.members]
        for i, (y, w) in enumerate(zip(ys, wss)):
            self._yp[i] = y.data_ptr()
            self._wp[i] = w.data_ptr()
        rc = self._fn(self.descs, self.layouts, len(ys), x.data_ptr(), self._yp, self._flags, self._wp, self._wb, sp)
        if rc == B.E_UNSUPPORTED:
            return None
        if rc:
            for w in wss:
                w.zero_()
            B.check(rc, "vptq_quant_gemv_sliced_grouped")
        return ys
