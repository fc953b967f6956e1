#!/usr/bin/env python3
"""Decode-GEMV benchmark for the VPTQ hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--hidden 8192] [--mode ...]

N = 1 (default): BASELINE.json configs[1] at the hidden size the north-star target is quoted on:
one VQuantLinear H x H, vector_len 8, k = 256 + 256 residual centroids (2 bits / weight),
enable_norm, batch = 1, seq = 1, fp16.  A *step* is one decode token through a RING of R
distinct layers (R x packed-index bytes >= 512 MiB, so neither the 32 MiB of L2 nor the 256 MiB
Infinity Cache can hold the weights between uses): R fused dequant + GEMV launches through the
C ABI, captured once in a hipGraph and replayed.  After W warm-up steps, `--regions` (5) timed
regions of EXACTLY K steps each are run, every one bracketed by a barrier + synchronize; the
line reports the MEDIAN region (all of them are listed under "regions_ms_per_step").
  --mode chain   : the ring as ONE persistent launch that walks its layers one after the other
                   (vptq_quant_gemv_chain, independent layers: the next layer's codebooks, activations
                   and first index words are requested while the current one streams)     [N = 1 default]
  --mode single  : one launch per layer (the reference's operator granularity)
  --mode chain_dep : the same launch with every layer reading the previous one's output (dependent chain)
  --mode grouped : layers launched 4 at a time with vptq_quant_gemv_grouped
  --mode tp_row  : BASELINE config #5: the linear projections of Llama-3-70B decoder layers
                   (q / o 8192^2, k / v 1024 x 8192, gate / up 28672 x 8192, down 8192 x 28672),
                   every projection cut row-parallel (input columns, vptq_amd.utils.shard.
                   shard_in_features) over the N ranks: fused GEMV with fp32 partial output ->
                   RCCL all-reduce over xGMI -> one rounding.  Strong scaling (total work fixed). [N > 1 default]
  --mode tp      : output rows split over the ranks, RCCL all-gather per layer (strong scaling)
  --mode rings   : every rank its own ring of independent layers, no collective (weak scaling);
                   with N > 1 this line is also attached to the tp_row line as "weak_scaling"
Extras (N = 1, unless --no-extras): one launch per layer, the dependent chain, hidden 4096 (BASELINE
configs[0/1]; chain and single), the reference's roundings (VPTQ_GEMV_EXACT), grouped x4, 16 tokens
(batched-decode kernel), the k = 8192 + 256 format (LDS-resident codebooks), tp_row on one GPU (the
strong-scaling baseline), each a short ring of its own; the prefill configuration (BASELINE configs[3]:
batch 4 x seq 2048 through 4096x4096 and 14336x4096, bf16, TFLOP/s of both routes) and the
Llama-3-8B shaped decode loop (BASELINE configs[2]: tokens/s + TTFT of the whole model,
tools/llama_decode.py in a process of its own).

Inputs and weights are resident in HBM before the timed region.  Prints ONE JSON line.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec (/opt/skills/guides/MI355X_MICROARCH.md)
LLAMA70B = dict(hidden=8192, ffn=28672, kv=1024)


def alg_bytes(I, O=None, k=256, kr=256, tokens=1):
    """Algorithmic bytes of one launch (DESIGN.md section 5 / SURVEY.md 8d): packed indices + both
    codebooks + x + scale + bias + y."""
    O = I if O is None else O
    T = int(np.log2(k)) + (int(np.log2(kr)) if kr > 0 else 0)
    return (O // 8) * ((I * T + 31) // 32) * 4 + (k + max(kr, 0)) * 8 * 2 + tokens * 2 * I + 4 * I + tokens * 2 * O


def model_decode_extra(timeout_s=240, arithmetic="reference"):
    """BASELINE configs[2]: the Llama-3-8B shaped decode loop of tools/llama_decode.py (all 32 decoder
    layers, every nn.Linear a 2-bit VQuantLinear through HF Transformers' VPTQ route, sibling projections
    in grouped launches, decode step replayed from a hipGraph) in a process of its own, after the timed
    regions of this one.  tokens/s and TTFT of the WHOLE model: HF's own eager kernels (norms, RoPE,
    SDPA, cache update) and the fp16 lm_head are in it, the VQuantLinear launches are about a quarter."""
    import subprocess
    try:
        p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "llama_decode.py"), "--fuse", "--new", "256"],
                           capture_output=True, text=True, timeout=timeout_s, env=dict(os.environ, VPTQ_ARITHMETIC=arithmetic))
        line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
        d = json.loads(line)
        return {"what": d["model"] + f", prompt 128, 256 new tokens, batch 1, one GPU; tools/llama_decode.py --fuse; arithmetic: {arithmetic}",
                "tokens_per_s": d["decode_tok_s_hipgraph"], "tokens_per_s_eager": d["decode_tok_s_eager"],
                "ttft_ms": d["ttft_ms"], "packed_index_GB": d["packed_index_GB"],
                "weight_GBps": d.get("hipgraph_weight_GBps"),
                "vqlinear_us_per_token": d.get("vqlinear_us_per_token"), "vqlinear_GBps": d.get("vqlinear_GBps"),
                "vqlinear_share_of_step": d.get("vqlinear_share_of_step")}
    except Exception as e:  # the headline line must not depend on transformers being importable
        return {"error": f"{type(e).__name__}: {e}"[:300]}


def prefill_extra(dev, tokens=8192):
    """BASELINE configs[3]: batch 4 x seq 2048 (8192 tokens) through one VQuantLinear 4096x4096 and one
    14336x4096, bf16, both routes: HIP dequant to a dense W + hipBLASLt GEMM (the reference's structure,
    vptq/ops/quant_gemm.py:231-274; the module's default) and the fused dequant-tile -> LDS -> MFMA GEMM
    (vptq_quant_gemm).  TFLOP/s = 2 x tokens x I x O / time; MFMA-busy figures come from the committed
    rocprofv3 PMC summary of the same shapes."""
    from vptq_amd import ops
    g = torch.Generator(device=dev).manual_seed(11)
    out = {"what": f"batch 4 x seq 2048 = {tokens} tokens, bf16, VQuantLinear v=8 k=256+256; dense = vptq_dequant + "
                   "hipBLASLt (default route), fused = vptq_quant_gemm (gemm_fused.hip, opt-in)", "shapes": {}}
    for (O, I) in ((4096, 4096), (14336, 4096)):
        m = make_layer(I, O, dev, g, dtype=torch.bfloat16)
        x = torch.randn(4, tokens // 4, I, device=dev, dtype=torch.bfloat16, generator=g)
        desc = m._descriptor()[1]

        def t_us(fn, iters=5):
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            ts = []
            for _ in range(iters):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); fn(); e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            return sorted(ts)[len(ts) // 2]
        with torch.no_grad():
            dense = t_us(lambda: m(x))
            fused = t_us(lambda: ops.quant_gemm_fused(x, desc, O))
            yd, yf = m(x), ops.quant_gemm_fused(x, desc, O)
        flops = 2.0 * tokens * I * O
        out["shapes"][f"{O}x{I}"] = {
            "dense_us": dense, "dense_TFLOPs": flops / dense / 1e6, "dense_frac_of_2500TF": flops / dense / 1e6 / 2500,
            "fused_us": fused, "fused_TFLOPs": flops / fused / 1e6, "fused_frac_of_2500TF": flops / fused / 1e6 / 2500,
            "fused_vs_dense_rel_diff": float(((yf.float() - yd.float()).abs().max() / yd.float().abs().max()).item())}
        del m, x, yd, yf
        torch.cuda.empty_cache()
    rounds = sorted(d for d in os.listdir(os.path.join(ROOT, "profiles")) if d[:1] == "r" and d[1:].isdigit())
    for rd in reversed(rounds):
        cand = os.path.join(ROOT, "profiles", rd, "prefill_bf16_pmc_summary.json")
        if os.path.exists(cand):
            out["mfma_busy"] = json.load(open(cand))
            out["mfma_busy_source"] = os.path.relpath(cand, ROOT)
            break
    return out


def k65536_extra(lib, B, dev, H, steps, warmup, regions, R=8, kr=256, v=8, light=False):
    """v8-k65536-256 (T = 24 bits, the format of most published checkpoints), H x H, one token: the library's default
    route (gemv_gather_kernel: centroid gathers through the caches) and the one-token GEMV over the load-time derived
    sliced layout (gemv_sliced.hip, VQuantLinear.enable_sliced_layout) on the same ring of R distinct layers."""
    from vptq_amd.utils.sliced import SlicedGemv
    g = torch.Generator(device=dev).manual_seed(4321)
    layers = [make_layer(H, H, dev, g, 65536, kr, v=v) for _ in range(R)]
    x = torch.randn(1, 1, H, device=dev, generator=torch.Generator(device=dev).manual_seed(7)).half()
    ys = [torch.empty(1, 1, H, device=dev, dtype=torch.float16) for _ in range(R)]
    descs = [layer_desc(m) for m in layers]
    # (light: only the product's default route - the exact sliced kernel where it serves the format, else the gather kernel)
    sls = [] if light else [SlicedGemv(m) for m in layers]
    # the reference's roundings over a layout (round 5; no residual codebook or the 256-entry one): what VQuantLinear.forward
    # takes for these layers in the default arithmetic
    exact_ok = bool(lib.vptq_sliced_layout_supported_for(descs[0][0], B.GEMV_EXACT))
    sxs = [SlicedGemv(m, exact=True) for m in layers] if exact_ok else []
    # round 6, opt-in selective arithmetic: the two-table formats over the FOLDED layouts with VPTQ_GEMV_SELECTIVE (gemv_hot.hip pre-pass)
    sel_ok = kr >= 4096 and bool(lib.vptq_quant_gemv_sliced_selective_supported(descs[0][0]))
    ssel = [SlicedGemv(m, selective=True) for m in layers] if sel_ok else []

    def default_pass():
        sp = torch.cuda.current_stream().cuda_stream
        for i in range(R):
            rc = lib.vptq_quant_gemv(descs[i][0], x.data_ptr(), ys[i].data_ptr(), 1, 0, None, 0, sp)
            assert rc == 0, lib.vptq_last_error()

    def sliced_pass():
        for i in range(R):
            sls[i](x, ys[i])

    def exact_sliced_pass():
        for i in range(R):
            sxs[i](x, ys[i])

    def selective_sliced_pass():
        for i in range(R):
            ssel[i](x, ys[i])
    T = 16 + (int(np.log2(kr)) if kr else 0)
    ab = (H // v) * ((H * T + 31) // 32) * 4 + (65536 + kr) * v * 2 + 2 * H + 4 * H + 2 * H
    out = {"what": f"VQuantLinear {H}x{H} v={v} k=65536+{kr} (T = {T} bits per index), ring of {R} layers; GB/s of the PACKED format's "
                   "algorithmic bytes for both routes (the sliced layouts read 4 bytes per element and table, 5 for k65536+256)"}
    routes = (("default", default_pass), ("sliced_layout", sliced_pass)) + ((("exact_sliced_layout", exact_sliced_pass),) if exact_ok else ())
    if light:
        routes = routes[2:] if exact_ok else routes[:1]
    if sel_ok:
        routes = routes + (("selective_sliced_layout", selective_sliced_pass),)
    for key, fn in routes:
        t = Timer(dev).run(fn, steps, warmup, regions)
        us = t["event_ms"] * 1e3 / (steps * R)
        out[key] = {"us_per_layer": us, "GBps": ab / us / 1e3, "frac_of_8TBps": ab / us / 1e3 / 8000.0}
    if "default" in out:
        out["default"]["kernel"] = lib.vptq_quant_gemv_kernel_name(descs[0][0], 1, 0).decode()
        out["default"]["what"] = "vptq_quant_gemv: centroid gathers through the caches (reference roundings); the module's route for layers the exact sliced kernel does not take"
    if "sliced_layout" in out:
        out["sliced_layout"]["kernel"] = "gemv_sliced_kernel"
        out["sliced_layout"]["what"] = "OPT-IN folded arithmetic over the load-time derived layouts"
    if sel_ok:
        out["selective_sliced_layout"]["kernel"] = "gemv_hot_kernel + gemv_sliced_kernel"
        out["selective_sliced_layout"]["what"] = ("OPT-IN selective arithmetic (round 6): pre-pass (threshold, hot blocks zeroed, their exact products) + the "
                                                  "folded launch over the layouts - what VQuantLinear.forward takes for two-table formats after "
                                                  "vptq_amd.set_arithmetic('selective')")
    if exact_ok:
        out["exact_sliced_layout"]["kernel"] = "gemv_sliced_kernel<EX>"
        out["exact_sliced_layout"]["what"] = ("the reference's roundings over a load-time derived layout (LDS-local gathers): the module's one-token route "
                                              "for this format in the default arithmetic")
        out["exact_sliced_layout"]["slices"] = sxs[0].slices
        exact_sliced_pass()
        torch.cuda.synchronize()
        got_x = ys[-1].clone()
    if not light:
        out["sliced_layout"]["layout_MiB_per_layer"] = sls[0].extra_bytes / 2**20
        out["sliced_layout"]["packed_index_MiB_per_layer"] = layers[0].indices.numel() * 4 / 2**20
        sliced_pass()
        torch.cuda.synchronize()
    layers[-1].enable_sliced_layout(False)   # (reference = the gather kernel, not the module's default one-token route)
    ref = layers[-1](x)
    if not light:
        out["sliced_vs_default_rel_diff"] = ((ys[-1].float() - ref.float()).abs().max() / ref.float().abs().max()).item()
    if exact_ok:
        out["exact_sliced_vs_default_bit_identical"] = float((got_x.view(torch.int16) == ref.view(torch.int16)).float().mean())
        assert out["exact_sliced_vs_default_bit_identical"] >= 0.95, out
    if light:
        del layers, sxs, ssel
        torch.cuda.empty_cache()
        return out
    # 2 and 4 tokens: the gather kernel (vptq_quant_gemv) against ONE launch over the same layouts (gemv_sliced_tok.hip:
    # column phases; 4 tokens contract on the matrix pipe), where the library takes the layer
    for tokens in (2, 4):
        if not all(sl.tokens_supported(tokens) for sl in sls):
            continue
        xt = torch.randn(1, tokens, H, device=dev, generator=torch.Generator(device=dev).manual_seed(8)).half()
        yt = [torch.empty(1, tokens, H, device=dev, dtype=torch.float16) for _ in range(R)]

        def gather_pass():
            sp = torch.cuda.current_stream().cuda_stream
            for i in range(R):
                rc = lib.vptq_quant_gemv(descs[i][0], xt.data_ptr(), yt[i].data_ptr(), tokens, 0, None, 0, sp)
                assert rc == 0, lib.vptq_last_error()

        def tokens_pass():
            for i in range(R):
                assert sls[i].forward_tokens(xt, yt[i]) is not None
        tokens_pass()
        torch.cuda.synchronize()
        got = yt[-1].clone()
        row = {}
        for key, fn in (("default", gather_pass), ("sliced_one_launch", tokens_pass)):
            t = Timer(dev).run(fn, steps, warmup, regions)
            row[key + "_us_per_layer"] = t["event_ms"] * 1e3 / (steps * R)
        gather_pass()
        torch.cuda.synchronize()
        row["rel_diff"] = ((got.float() - yt[-1].float()).abs().max() / yt[-1].float().abs().max()).item()
        assert row["rel_diff"] <= 2e-3, row     # (both within 1e-3 of the reference)
        out[f"tokens{tokens}"] = row
    # round 5: 2 and 3 tokens in the reference's roundings - the gather kernel (VPTQ_GEMV_EXACT) against ONE PASS of the exact sliced
    # kernel over the exact layout (gemv_sliced<EX, TOK>: every token's activations beside the slice, the weight rebuilt once; two
    # tables of v = 8: the residual entries gathered once) - what VQuantLinear.forward takes for these layers in the default arithmetic
    try:
        for tokens in (2, 3):
            if not sxs or not all(sx.tokens_one_pass(tokens) for sx in sxs):
                continue
            xt = torch.randn(1, tokens, H, device=dev, generator=torch.Generator(device=dev).manual_seed(9)).half()
            yt = [torch.empty(1, tokens, H, device=dev, dtype=torch.float16) for _ in range(R)]

            def gather_exact_pass():
                sp = torch.cuda.current_stream().cuda_stream
                for i in range(R):
                    rc = lib.vptq_quant_gemv(descs[i][0], xt.data_ptr(), yt[i].data_ptr(), tokens, B.GEMV_EXACT, None, 0, sp)
                    assert rc == 0, lib.vptq_last_error()

            def one_pass():
                for i in range(R):
                    assert sxs[i].forward_tokens(xt, yt[i]) is not None
            one_pass()
            torch.cuda.synchronize()
            got = yt[-1].clone()
            row = {"kernel": f"gemv_sliced_kernel<EX, TOK = {tokens}>"}
            for key, fn in (("default", gather_exact_pass), ("exact_sliced_one_pass", one_pass)):
                t = Timer(dev).run(fn, steps, warmup, regions)
                row[key + "_us_per_layer"] = t["event_ms"] * 1e3 / (steps * R)
            gather_exact_pass()
            torch.cuda.synchronize()
            row["bit_identical_to_default"] = float((got.view(torch.int16) == yt[-1].view(torch.int16)).float().mean())
            assert row["bit_identical_to_default"] >= 0.95, row
            out[f"exact_tokens{tokens}"] = row
    except Exception as e:   # (an extra of an extra)
        out["exact_tokens_error"] = f"{type(e).__name__}: {e}"[:300]
    del layers, sls, sxs, ssel
    torch.cuda.empty_cache()
    return out


def wide_layer_extra(lib, B, dev, steps, warmup, regions, I=28672, O=8192, R=4, kr=256):
    """a layer too wide for the reference's roundings in one piece (28672 columns: the down projections of the 70B class), one
    token: the gather kernel against the exact sliced kernel over COLUMN PARTS (2 x 14336 columns, one grouped launch, shared
    accumulator words: VPTQ_GEMV_COLUMN_PARTS) - what VQuantLinear.forward takes for it in the default arithmetic (round 5)"""
    from vptq_amd.utils.sliced import SlicedGemv
    g = torch.Generator(device=dev).manual_seed(4322)
    layers = [make_layer(I, O, dev, g, 65536, kr) for _ in range(R)]
    x = torch.randn(1, 1, I, device=dev, generator=torch.Generator(device=dev).manual_seed(10)).half()
    ys = [torch.empty(1, 1, O, device=dev, dtype=torch.float16) for _ in range(R)]
    descs = [layer_desc(m) for m in layers]
    sxs = [SlicedGemv(m, exact=True) for m in layers]

    def default_pass():
        sp = torch.cuda.current_stream().cuda_stream
        for i in range(R):
            rc = lib.vptq_quant_gemv(descs[i][0], x.data_ptr(), ys[i].data_ptr(), 1, B.GEMV_EXACT, None, 0, sp)
            assert rc == 0, lib.vptq_last_error()

    def parts_pass():
        for i in range(R):
            assert sxs[i](x, ys[i]) is not None
    T = 16 + (int(np.log2(kr)) if kr else 0)
    ab = (O // 8) * ((I * T + 31) // 32) * 4 + (65536 + kr) * 8 * 2 + 2 * I + 4 * I + 2 * O
    out = {"what": f"VQuantLinear {I}x{O} v=8 k=65536+{kr}, ring of {R} layers, one token, reference roundings; GB/s of the PACKED format's algorithmic bytes",
           "column_parts": sxs[0].parts, "slices": sxs[0].slices}
    parts_pass()
    torch.cuda.synchronize()
    got = ys[-1].clone()
    for key, fn in (("default", default_pass), ("exact_sliced_column_parts", parts_pass)):
        t = Timer(dev).run(fn, steps, warmup, regions)
        us = t["event_ms"] * 1e3 / (steps * R)
        out[key] = {"us_per_layer": us, "GBps": ab / us / 1e3, "frac_of_8TBps": ab / us / 1e3 / 8000.0}
    out["default"]["kernel"] = lib.vptq_quant_gemv_kernel_name(descs[0][0], 1, B.GEMV_EXACT).decode()
    out["exact_sliced_column_parts"]["kernel"] = "gemv_sliced_kernel<EX>"
    default_pass()
    torch.cuda.synchronize()
    out["bit_identical_to_default"] = float((got.view(torch.int16) == ys[-1].view(torch.int16)).float().mean())
    assert out["bit_identical_to_default"] >= 0.95, out
    del layers, sxs
    torch.cuda.empty_cache()
    return out


def sysfs_card_of_device(dev_index=0):
    """/sys/class/drm/cardN/device of HIP device `dev_index`, matched by PCI address (a box of the pool shows every GPU
    of its node in /sys; the first card is not necessarily the one this process runs on).  None if it cannot be told."""
    import glob
    try:
        bus = torch.cuda.get_device_properties(dev_index).pci_bus_id
        dom = getattr(torch.cuda.get_device_properties(dev_index), "pci_domain_id", 0)
        devid = getattr(torch.cuda.get_device_properties(dev_index), "pci_device_id", 0)
        want = f"{dom:04x}:{bus:02x}:{devid:02x}."
    except Exception:
        return None
    for c in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
        if os.path.basename(os.path.realpath(c)).lower().startswith(want):
            return c
    return None


class SclkSampler:
    """Shader clock of this GPU during the timed regions (rule DVFS in or out as the box-to-box
    spread): a thread reads the current level of pp_dpm_sclk every 20 ms.  None where sysfs does not offer it."""

    def __init__(self, dev_index=0):
        import glob
        # the measured shader clock (hwmon freq1_input, Hz) where the driver offers it; else the DPM level table
        # (pp_dpm_sclk: the starred level - on MI300-class parts that is a coarse state, not the live clock)
        self.card = sysfs_card_of_device(dev_index)   # (matched by PCI address: the node's other GPUs are in /sys too)
        hw = sorted(glob.glob(os.path.join(self.card, "hwmon", "hwmon*", "freq1_input"))) if self.card else []
        self.path = hw[0] if hw else (os.path.join(self.card, "pp_dpm_sclk") if self.card else None)
        self.kind = "hwmon freq1_input" if hw else "pp_dpm_sclk (starred level)"
        self.samples, self._stop, self._thread = [], False, None
        # package power (hwmon power1_input, microwatts; label PPT) and its cap, where the driver offers them
        pw = sorted(glob.glob(os.path.join(self.card, "hwmon", "hwmon*", "power1_input"))) if self.card else []
        self.power_path = pw[0] if pw else None
        self.power_samples = []
        self.power_cap_w = None
        # memory clock (hwmon freq2_input, Hz) where offered: boxes of the pool differ in what the memory system delivers
        mc = sorted(glob.glob(os.path.join(self.card, "hwmon", "hwmon*", "freq2_input"))) if self.card else []
        self.mclk_path = mc[0] if mc else None
        self.mclk_samples = []
        if pw:
            try:
                self.power_cap_w = float(open(os.path.join(os.path.dirname(pw[0]), "power1_cap")).read().strip()) / 1e6
            except Exception:
                pass

    def _read(self):
        try:
            if self.kind.startswith("hwmon"):
                return float(open(self.path).read().strip()) / 1e6
            for line in open(self.path):
                if line.rstrip().endswith("*"):
                    return float(line.split(":")[1].lower().replace("mhz", "").replace("*", "").strip())
        except Exception:
            return None
        return None

    def __enter__(self):
        if self.path is not None:
            import threading

            def loop():
                while not self._stop:
                    v = self._read()
                    if v is not None:
                        self.samples.append(v)
                    if self.power_path is not None:
                        try:
                            self.power_samples.append(float(open(self.power_path).read().strip()) / 1e6)
                        except Exception:
                            pass
                    if self.mclk_path is not None:
                        try:
                            self.mclk_samples.append(float(open(self.mclk_path).read().strip()) / 1e6)
                        except Exception:
                            pass
                    time.sleep(0.02)
            self._thread = threading.Thread(target=loop, daemon=True)
            self._thread.start()
        return self

    def __exit__(self, *exc):
        self._stop = True
        if self._thread is not None:
            self._thread.join(timeout=1.0)

    def summary(self):
        if not self.samples:
            return None
        s = sorted(self.samples)
        return {"samples": len(s), "min_mhz": s[0], "median_mhz": s[len(s) // 2], "max_mhz": s[-1],
                "source": f"{self.kind} of the card whose PCI address is this HIP device's, every 20 ms (before round 3's "
                          "last session the first card of /sys was read - on a box of the pool that is another GPU of the node)"}

    def mclk_mhz(self):
        s = sorted(self.mclk_samples)
        return s[len(s) // 2] if s else None

    def power_summary(self):
        if not self.power_samples:
            return None
        s = sorted(self.power_samples)
        return {"samples": len(s), "median_w": s[len(s) // 2], "max_w": s[-1], "min_w": s[0], "cap_w": self.power_cap_w,
                "source": "hwmon power1_input (PPT, package power) of the same card, every 20 ms"}


def shard_ring(total_layers, rank, world):
    """Layer-parallel sharding: independent VQuantLinear layers are dealt round-robin to
    ranks; no rank ever needs another rank's layer (no data-path collective)."""
    return list(range(rank, total_layers, world))


def reduce_times(wall_s, event_ms, dist_mod=None, device="cpu"):
    """max over ranks of (wall seconds, HIP-event milliseconds)."""
    if dist_mod is None:
        return wall_s, event_ms
    tt = torch.tensor([wall_s, event_ms], device=device, dtype=torch.float64)
    dist_mod.all_reduce(tt, op=dist_mod.ReduceOp.MAX)
    return tuple(tt.tolist())


def job_throughput_gbps(world, bytes_per_launch_layer, layers_per_rank, steps, wall_s):
    """Whole-job GB/s: every rank streams its own ring `steps` times in `wall_s`."""
    return world * bytes_per_launch_layer * layers_per_rank * steps / wall_s / 1e9


def make_layer(I, O, dev, g, k=256, kr=256, dtype=torch.float16, v=8):
    import vptq_amd
    m = vptq_amd.VQuantLinear(
        I, O, vector_lens=[-1, v], num_centroids=[-1, k], num_res_centroids=[-1, kr if kr > 0 else -1],
        group_num=1, group_size=I, outlier_size=0, indices_as_float=False, enable_norm=True,
        enable_perm=False, is_indice_packed=True, bias=False, dtype=dtype,
        device=dev, enable_proxy_error=False)
    m.indices.data = torch.randint(-2**31, 2**31 - 1, m.indices.shape, generator=g,
                                   device=dev, dtype=torch.int64).to(torch.int32)
    m.centroids.weight.data = (torch.randn(m.centroids.weight.shape, generator=g, device=dev) * 0.02).to(dtype)
    if kr > 0:
        m.res_centroids.weight.data = (torch.randn(m.res_centroids.weight.shape, generator=g, device=dev) * 0.005).to(dtype)
    m.weight_scale.data = (1 + 0.1 * torch.randn(I, generator=g, device=dev)).to(dtype)
    m.weight_bias.data = (0.01 * torch.randn(I, generator=g, device=dev)).to(dtype)
    return m.eval()


def layer_desc(m):
    """(C-ABI descriptor, keep-alive) of a vptq_amd.VQuantLinear: the module's own cached descriptor"""
    c = m._descriptor()
    return c[1], (m, c[2])


def make_ring(H, R, dev, seed, k=256, kr=256, dtype=torch.float16):
    g = torch.Generator(device=dev).manual_seed(seed)
    return [make_layer(H, H, dev, g, k, kr, dtype=dtype) for _ in range(R)]


def layer_spec(layer):
    """vptq_amd.VQuantLinear (canonical-style, one codebook) -> oracle LayerSpec"""
    from oracle import vptq_oracle as vo
    to_np = lambda t: t.detach().cpu().contiguous().view(torch.int16).numpy().view(np.uint16)  # noqa: E731
    k, kr = layer.num_centroids, layer.num_res_centroids
    L = vo.LayerSpec(layer.in_features, layer.out_features, 8, k, kr, 1, layer.in_features, dtype="f16")
    L.indices = layer.indices.detach().cpu().numpy()
    L.centroids = to_np(layer.centroids.weight).reshape(1, k, 8)
    L.res_centroids = to_np(layer.res_centroids.weight).reshape(1, kr, 8)
    L.weight_scale, L.weight_bias = to_np(layer.weight_scale), to_np(layer.weight_bias)
    return L


def rel_err_bits(y_gpu, y_ref_bits):
    from oracle import vptq_oracle as vo
    a = y_gpu.detach().float().cpu().numpy().reshape(-1).astype(np.float64)
    b = vo.to_f32(np.asarray(y_ref_bits).reshape(-1), "f16").astype(np.float64)
    return float(np.abs(a - b).max() / np.abs(b).max())


def cpu_baseline(layer, x, y_gpu, H):
    """The reference's CPU path on THIS box's host cores, one H x H forward, median of 3:
    * `cpu_baseline`: the torch restatement of the reference's own tensor-op sequence
      (oracle/torch_ref.py: one-element-per-bit unpack, gathers, add, scale, bias, F.linear -
      vptq/ops/quant_gemm.py:43-158, 231-274; pinned on the reference's goldens), torch's thread
      pool pinned to the core count it reports;
    * `cpu_baseline_c_port`: the C oracle (oracle/vptq_oracle.c, OpenMP) - a fairer, leaner CPU
      implementation of the same arithmetic; ours, not the reference's.
    Also returns the parity error of the GPU result against the C oracle."""
    from oracle import c_oracle as co
    from oracle import torch_ref as tr
    ab = alg_bytes(H)
    out = {}
    # ---- torch restatement, at the thread count that is FASTEST on this box (the reference's CPU path as
    # well as this host runs it - an over-subscribed pool is not the reference's speed)
    cores = os.cpu_count() or 1
    cpu = lambda t: None if t is None else t.detach().cpu()  # noqa: E731
    args = (cpu(layer.indices), cpu(layer.centroids.weight), cpu(layer.res_centroids.weight),
            cpu(layer.weight_scale), cpu(layer.weight_bias))
    kw = dict(num_centroids=256, num_res_centroids=256, vector_len=8, group_size=H, out_features=H)
    xc = cpu(x)
    sweep = {}
    with torch.no_grad():
        torch.set_num_threads(min(cores, 32))
        y_t = tr.forward(xc, *args, **kw)      # warm-up (page-in)
        for n in sorted({n for n in (8, 16, 32, 64, cores) if n <= cores}):
            torch.set_num_threads(n)
            t0 = time.perf_counter()
            y_t = tr.forward(xc, *args, **kw)
            sweep[n] = time.perf_counter() - t0
        best = min(sweep, key=sweep.get)
        torch.set_num_threads(best)
        ts = [sweep[best]]
        for _ in range(2):
            t0 = time.perf_counter()
            y_t = tr.forward(xc, *args, **kw)
            ts.append(time.perf_counter() - t0)
    t = sorted(ts)[1]
    out["cpu_baseline"] = {
        "value": ab / t / 1e9, "unit": "GB/s", "cores": int(best), "kind": "port",
        "sample": f"1 VQuantLinear {H}x{H} forward, torch restatement of the reference's CPU tensor-op "
                  f"sequence (oracle/torch_ref.py), {best} torch threads = the fastest of "
                  f"{ {n: round(v, 2) for n, v in sweep.items()} } s per forward (os.cpu_count() = {cores}), "
                  f"median of 3 at that count: {t * 1e3:.0f} ms"}
    torch.set_num_threads(min(cores, 32))
    rel_t = float(((y_gpu.detach().float().cpu() - y_t.float()).abs().max() / y_t.float().abs().max()).item())
    out["parity_rel_err_vs_torch_restatement"] = rel_t
    # ---- C port
    rel = None
    if co.available():
        L = layer_spec(layer)
        xb = xc.contiguous().view(torch.int16).numpy().view(np.uint16)
        scratch = np.empty((H, H), dtype=np.uint16)
        threads = co.lib().vo_num_threads()
        y = co.forward(L, xb, scratch=scratch)          # warm-up (page-in)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            y = co.forward(L, xb, scratch=scratch)
            ts.append(time.perf_counter() - t0)
        t = sorted(ts)[1]
        rel = rel_err_bits(y_gpu, y)
        out["cpu_baseline_c_port"] = {
            "value": ab / t / 1e9, "unit": "GB/s", "cores": int(threads), "kind": "port",
            "sample": f"same forward, C oracle (oracle/vptq_oracle.c: dequant to dense W + linear), "
                      f"OpenMP {threads} threads, median of 3, {t * 1e3:.0f} ms each"}
        out["parity_rel_err_vs_cpu_oracle"] = rel
    return out, (rel if rel is not None else rel_t)


def compute_floor(kname, H):
    """What bounds the kernel besides HBM (tools/ubench*.hip on MI355X, profiles/r02/): per launch of
    one HxH layer, with every CU busy."""
    n_idx = (H // 8) * H                       # index pairs = weight vectors of 8
    if kname.startswith("gemv_k256m"):
        return {"what": "instruction issue on the SIMD's VALU / MFMA port: per index 4 MFMA 4x4x4 (8 cycles each) + 4 "
                        "v_perm_b32 + 2 ds_read_b128 ~ 60 cycles per index-wave at the ~1.9 GHz the CUs hold in this "
                        "kernel (trace build, s_memtime); timing-only ablations show the costs ADD, little hides behind "
                        "anything else (profiles/r02/k256m_ablation_8192.txt: no MFMAs -1.5 us per launch, no gathers "
                        "-1.0, neither -2.3); a launch needs 128 index-waves per SIMD at 8192^2.  Beside it: what a "
                        "launch that ONLY streams these bytes costs (tools/ubench_stream.hip, "
                        "profiles/r02/ubench_stream_8192.txt: 4.1-5.0 us per 16 MiB launch incl. the 1.8 us boundary)",
                "us_per_launch_issue": n_idx / 64 / 1024 * 60 / 1.9e3,
                "us_per_launch_pure_stream": 4.65}
    instr = 14 if "fast" in kname else 22
    return {"what": f"VALU issue: {instr} instructions per index, 2.25 ns per wave-instruction per "
                    "SIMD at 4 waves/SIMD (1024 SIMDs)",
            "us_per_launch": n_idx * instr / (1024 * 64) * 2.25e-3}


class Timer:
    """hipGraph capture of `one_pass`, W warm-up replays, then `regions` timed regions of exactly
    `steps` replays, each bracketed by a barrier + synchronize; HIP events on the launch stream;
    max over ranks."""

    def __init__(self, dev, dist=None, allow_eager=False):
        self.dev, self.dist, self.allow_eager = dev, dist, allow_eager
        self.stream = torch.cuda.Stream(device=dev)

    def soak(self, seconds):
        """the last workload again, back to back for `seconds` (power / clock readings need a load that lasts longer
        than the timed regions); returns microseconds per step"""
        n = 0
        with torch.cuda.stream(self.stream):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < seconds:
                for _ in range(20):
                    self.last_graph.replay()
                n += 20 * getattr(self, "last_steps_per_graph", 1)
                torch.cuda.synchronize()
            return (time.perf_counter() - t0) * 1e6 / n

    # Steps per captured graph: one graph launch costs ~8 us of host / front-end time between two replays
    # (profiles/r03: 150.6 us per replay by events against 142.4 us kernel-only), which belongs to the replay mechanism,
    # not to the step.  A timed region is still EXACTLY `steps` steps: steps / p replays of a graph holding p steps.
    MAX_STEPS_PER_GRAPH = 10
    # Conditioning: the shader clock of an idle MI355X sits at ~160 MHz and the package takes ~0.5 s under load to
    # settle at its power limit (profiles/r04/ubench_energy_table.txt; bursts of a few ms run up to 10 % faster than
    # the sustained rate, a cold start slower).  The workload is replayed for this long BEFORE the W warm-up steps, so
    # that the K timed steps measure the sustained state, whatever K and W are.  Reported as `conditioning_s`.
    CONDITIONING_S = float(os.environ.get("VPTQ_BENCH_CONDITIONING_S", "1.0"))

    def run(self, one_pass, steps, warmup, regions):
        dist = self.dist

        class _Eager:  # same interface as a captured graph
            def replay(self):
                one_pass()

        p = max(d for d in range(1, max(1, self.MAX_STEPS_PER_GRAPH) + 1) if steps % d == 0)
        res = []
        with torch.cuda.stream(self.stream):
            one_pass()
            torch.cuda.synchronize()
            captured = True
            # gloo collectives (VPTQ_BENCH_SAME_GPU test runs) cannot be captured, and a refused capture
            # leaves this ROCm runtime unusable (the recovery below segfaulted in 2 of 2 trials): do not try.
            # VPTQ_BENCH_NO_GRAPH=1 forces eager launches for a backend that turns out to refuse as well.
            no_graph = os.environ.get("VPTQ_BENCH_NO_GRAPH") == "1" or \
                (dist is not None and dist.get_backend() == "gloo")
            try:
                if no_graph and self.allow_eager:
                    raise RuntimeError("capture skipped")
                graph1 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph1, stream=self.stream):
                    one_pass()
                graph = graph1
                if p > 1:
                    graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(graph, stream=self.stream):
                        for _ in range(p):
                            one_pass()
            except Exception as e:
                if not self.allow_eager:
                    raise
                captured, graph, p = False, _Eager(), 1  # collectives that refuse capture: eager launches
                graph1 = graph
                print(f"[bench] hipGraph capture refused ({type(e).__name__}); eager launches", file=sys.stderr, flush=True)
        if not captured and not no_graph:
            # The failed capture leaves (a) its error as the runtime's "last error" - the next launch check,
            # this library's and torch's alike, would report it - and (b) the capture stream invalidated:
            # consume the one, continue on a fresh stream.
            import ctypes
            hip = ctypes.CDLL("libamdhip64.so")
            for _ in range(4):
                if hip.hipGetLastError() == 0:
                    break
            try:
                torch.cuda.synchronize()
            except Exception:
                pass
            hip.hipGetLastError()
            self.stream = torch.cuda.Stream(device=self.dev)
        with torch.cuda.stream(self.stream):
            cond = self.CONDITIONING_S if (captured and dist is None) else 0.0
            if cond > 0:
                t0 = time.perf_counter()
                while time.perf_counter() - t0 < cond:
                    for _ in range(20):
                        graph.replay()
                    torch.cuda.synchronize()
            for _ in range(warmup):
                graph1.replay()
            for _ in range(regions):
                torch.cuda.synchronize()
                if dist is not None:
                    dist.barrier()
                torch.cuda.synchronize()
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                t0 = time.perf_counter()
                e0.record(self.stream)
                for _ in range(steps // p):
                    graph.replay()
                e1.record(self.stream)
                torch.cuda.synchronize()
                if dist is not None:
                    dist.barrier()
                torch.cuda.synchronize()
                wall = time.perf_counter() - t0
                res.append(reduce_times(wall, e0.elapsed_time(e1), dist, self.dev))
            # the same K steps once more as K replays of the ONE-step graph (the rounds 1 - 3 methodology: every step pays the
            # ~8 us between two graph launches), so that numbers stay comparable across rounds; never the headline
            one_step_ms = None
            if captured and p > 1 and dist is None:
                torch.cuda.synchronize()
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                e0.record(self.stream)
                for _ in range(steps):
                    graph1.replay()
                e1.record(self.stream)
                torch.cuda.synchronize()
                one_step_ms = e0.elapsed_time(e1) / steps
        self.last_graph = graph
        self.last_steps_per_graph = p
        res.sort(key=lambda r: r[0])
        med = res[len(res) // 2]
        return dict(wall_s=med[0], event_ms=med[1], captured=captured, steps_per_graph=p, conditioning_s=cond,
                    regions_ms_per_step=[r[0] * 1e3 / steps for r in res], ms_per_step_one_step_per_graph=one_step_ms)


def bench_ring(lib, B, dev, timer, H, mode, flags, steps, warmup, regions, rank=0, world=1, group=4,
               tokens=1, k=256, kr=256, ring=0, tp_split=None, prefetch=False, chain=32, dtype=torch.float16):
    """One ring measurement; returns (result dict, layers, x, ys).  mode: single | grouped | chain | chain_dep"""
    T = int(np.log2(k)) + (int(np.log2(kr)) if kr > 0 else 0)
    idx_bytes = (H // 8) * (H * T // 32) * 4
    R = ring or max(2, (512 << 20) // idx_bytes)
    layers = make_ring(H, R, dev, seed=1234 + (0 if tp_split else rank), k=k, kr=kr, dtype=dtype)
    if tp_split == "out":
        from vptq_amd.utils.shard import shard_out_features
        layers = [shard_out_features(m, rank, world) for m in layers]
        torch.cuda.empty_cache()
    x = torch.randn(1, tokens, H, device=dev, generator=torch.Generator(device=dev).manual_seed(7)).to(dtype)
    ys = [torch.empty(1, tokens, layers[i].out_features, device=dev, dtype=dtype) for i in range(R)]
    y_full = torch.empty(H, device=dev, dtype=torch.float16) if tp_split == "out" else None
    descs, keeps = [], []
    if prefetch:
        from vptq_amd.layers.vqlinear import chain_prefetch
        chain_prefetch(layers, circular=True)
    for i, m in enumerate(layers):
        d, kk = layer_desc(m)
        descs.append(d)
        keeps.append(kk)
    kname = lib.vptq_quant_gemv_kernel_name(descs[0], tokens, flags).decode()
    dist = timer.dist
    if mode == "grouped":
        chunks = []
        for i0 in range(0, R, group):
            m = min(group, R - i0)
            chunks.append((m, (B.LayerDesc * m)(*descs[i0:i0 + m]),
                           (C.c_void_p * m)(*[x.data_ptr()] * m),
                           (C.c_void_p * m)(*[t.data_ptr() for t in ys[i0:i0 + m]])))
        launches = len(chunks)
        kname = lib.vptq_quant_gemv_grouped_kernel_name(chunks[0][1], chunks[0][0], tokens, flags).decode()

        def one_pass():
            sp = torch.cuda.current_stream().cuda_stream
            for m, arr, xp, yp in chunks:
                rc = lib.vptq_quant_gemv_grouped(arr, m, xp, yp, tokens, flags, sp)
                assert rc == 0, lib.vptq_last_error()
    elif mode in ("chain", "chain_dep"):
        # the ring as launches of `chain` layers each (32 = the most one launch takes); dependent: layer
        # i + 1 reads layer i's output
        dep = mode == "chain_dep"
        cflags = flags | (B.GEMV_CHAIN_DEPENDENT if dep else 0)
        per = R if dep else min(chain, 32, R)      # layers per call (a dependent chain is ONE call; the
        chunks = []                                # library cuts calls of more than 32 layers itself)
        for i0 in range(0, R, per):
            m = min(per, R - i0)
            chunks.append((m, (B.LayerDesc * m)(*descs[i0:i0 + m]),
                           (C.c_void_p * m)(*[(ys[i - 1] if dep and i > 0 else x).data_ptr() for i in range(i0, i0 + m)]),
                           (C.c_void_p * m)(*[t.data_ptr() for t in ys[i0:i0 + m]])))
        # arrival flags of a dependent chain / thresholds + corrections of VPTQ_GEMV_SELECTIVE; the calls follow each other on
        # one stream and share the buffer
        wsb = max(lib.vptq_quant_gemv_chain_workspace_bytes_for(arr, m, cflags) for m, arr, *_ in chunks)
        ws = torch.zeros(max(wsb, 4) // 4, dtype=torch.int32, device=dev) if wsb else None
        launches = sum((m + 31) // 32 for m, *_ in chunks)
        kname = lib.vptq_quant_gemv_chain_kernel_name(chunks[0][1], chunks[0][0], tokens, cflags).decode()
        keeps.append((chunks, ws))

        def one_pass():
            sp = torch.cuda.current_stream().cuda_stream
            for m, arr, xp, yp in chunks:
                rc = lib.vptq_quant_gemv_chain(arr, m, xp, yp, tokens, cflags, None if ws is None else ws.data_ptr(), wsb, sp)
                assert rc == 0, lib.vptq_last_error()
    else:
        launches = R
        # (scratch memory where the library's route for this call wants it: the one-pass batched-decode kernel)
        gwsb = lib.vptq_quant_gemv_workspace_bytes(descs[0], tokens, flags)
        gws = torch.empty(max(gwsb, 16), dtype=torch.uint8, device=dev)
        keeps.append(gws)

        def one_pass():
            sp = torch.cuda.current_stream().cuda_stream
            for i in range(R):
                rc = lib.vptq_quant_gemv(descs[i], x.data_ptr(), ys[i].data_ptr(), tokens, flags,
                                         gws.data_ptr() if gwsb else None, gwsb, sp)
                assert rc == 0, lib.vptq_last_error()
                if tp_split == "out" and dist is not None:
                    dist.all_gather_into_tensor(y_full, ys[i].view(-1))
    t = timer.run(one_pass, steps, warmup, regions)
    ab = alg_bytes(H, H, k, kr, tokens)
    strong = tp_split is not None
    value = job_throughput_gbps(1 if strong else world, ab, R, steps, t["wall_s"])
    us_per_launch = t["event_ms"] * 1e3 / (steps * launches)
    bytes_per_launch = ab * R / launches / (world if strong else 1)
    res = dict(value=value, ms_per_step=t["wall_s"] * 1e3 / steps, us_per_launch=us_per_launch,
               bytes_per_launch=bytes_per_launch, achieved=bytes_per_launch / us_per_launch / 1e3,
               kernel=kname, ring=R, launches_per_step=launches, hipgraph=t["captured"],
               us_per_layer=t["event_ms"] * 1e3 / (steps * R),
               regions_ms_per_step=t["regions_ms_per_step"], idx_mib=R * idx_bytes >> 20,
               steps_per_graph=t["steps_per_graph"], conditioning_s=t["conditioning_s"],
               ms_per_step_one_step_per_graph=t["ms_per_step_one_step_per_graph"])
    return res, layers, x, ys, keeps


def llama70b_projections():
    H, Fd, KV = LLAMA70B["hidden"], LLAMA70B["ffn"], LLAMA70B["kv"]
    return [("q_proj", H, H), ("k_proj", KV, H), ("v_proj", KV, H), ("o_proj", H, H),
            ("gate_proj", Fd, H), ("up_proj", Fd, H), ("down_proj", H, Fd)]   # (name, O, I)


def bench_tp_row(lib, B, dev, timer, rank, world, n_layers, flags, steps, warmup, regions, check=True,
                 fuse_siblings=True, pair=False):
    """BASELINE config #5 (see the module docstring).  Every rank builds the SAME full projections
    (common seed; the permutation is already absorbed: enable_perm = False) and keeps its slice of
    the input columns; per projection: fused GEMV of the slice with fp32 partial output
    (VPTQ_GEMV_OUT_F32) -> all-reduce (RCCL over xGMI) -> one rounding to fp16.
    fuse_siblings (default): q / k / v and gate / up - projections of the same input - go out as ONE
    grouped launch (vptq_quant_gemv_grouped) into one contiguous fp32 buffer, ONE all-reduce and one
    rounding each: 4 launches + 4 collectives per decoder layer instead of 7 + 7 (the collectives are
    latency-bound at batch 1: fewer, larger ones).
    pair (--mode tp_pair): the Megatron pairing SURVEY 8(e) describes - q / k / v / gate / up cut
    COLUMN-parallel (output rows, shard_out_features: disjoint output slices, no collective; their
    consumers o / down are row-parallel over the same split), o / down row-parallel as above: 2
    all-reduces per decoder layer instead of 4."""
    from vptq_amd.utils.shard import shard_in_features, shard_out_features
    col_names = ("q_proj", "k_proj", "v_proj", "gate_proj", "up_proj") if pair else ()
    dist = timer.dist
    projs = llama70b_projections()
    groups = [[0, 1, 2], [3], [4, 5], [6]] if fuse_siblings else [[i] for i in range(len(projs))]
    g = torch.Generator(device=dev).manual_seed(4321)
    check_names = ("q_proj", "k_proj", "gate_proj", "down_proj")   # one projection of EVERY shape, layer 0
    calls, keeps, full0, checks = [], [], {}, []
    for li in range(n_layers):
        mods = []
        for (name, O, I) in projs:
            m = make_layer(I, O, dev, g)
            if li == 0 and name in check_names and check and rank == 0:
                full0[name] = m                # kept whole for the parity check
            if name in col_names:
                mods.append((shard_out_features(m, rank, world), O, I))
            else:
                mods.append((shard_in_features(m, rank, world), O, I))
            del m
        for grp in groups:
            I = mods[grp[0]][2]
            x = torch.randn(1, 1, I, device=dev, dtype=torch.float16, generator=g)   # siblings share their input
            s0 = mods[grp[0]][0]
            col = s0.shard[0] == "out"
            xs = x if col else x[..., s0.shard[1]:s0.shard[2]].contiguous()
            Os = [mods[i][0].out_features for i in grp]   # (column-parallel: this rank's slice of the rows)
            part = None if col else torch.empty(1, 1, sum(Os), device=dev, dtype=torch.float32)
            y = torch.empty(1, 1, sum(Os), device=dev, dtype=torch.float16)
            ds = []
            for i in grp:
                d, kk = layer_desc(mods[i][0])
                ds.append(d); keeps.append((kk, mods[i][0]))
            offs = [sum(Os[:j]) for j in range(len(grp))]
            arr = (B.LayerDesc * len(grp))(*ds)
            xp = (C.c_void_p * len(grp))(*[xs.data_ptr()] * len(grp))
            yp = (C.c_void_p * len(grp))(*[(y.data_ptr() + 2 * o) if col else (part.data_ptr() + 4 * o) for o in offs])
            calls.append((len(grp), arr, xp, yp, part, y))
            keeps.append((xs, part, y))
            if li == 0:
                for pi in grp:
                    if projs[pi][0] in full0:
                        o = offs[grp.index(pi)]
                        sh = mods[pi][0].shard
                        checks.append((projs[pi][0], x, y[..., o:o + Os[grp.index(pi)]],
                                       (sh[1], sh[2]) if col else None))
        del mods
        torch.cuda.empty_cache()
    fl = flags | B.GEMV_OUT_F32

    def one_pass():
        sp = torch.cuda.current_stream().cuda_stream
        for (m, arr, xp, yp, part, y) in calls:
            f = flags if part is None else fl   # column-parallel: this rank's rows, rounded in the kernel
            if m == 1:
                rc = lib.vptq_quant_gemv(arr, xp[0], yp[0], 1, f, None, 0, sp)
            else:
                rc = lib.vptq_quant_gemv_grouped(arr, m, xp, yp, 1, f, sp)
            assert rc == 0, lib.vptq_last_error()
            if part is None:
                continue
            if dist is not None:
                dist.all_reduce(part)           # fp32 partial sums over the ranks
            y.copy_(part)                       # the ONE rounding (reference: F.linear's)
    t = timer.run(one_pass, steps, warmup, regions)
    ab = sum(alg_bytes(I, O) for (_, O, I) in projs) * n_layers
    per_layer = len(groups)
    n_reduce = sum(1 for grp in groups if projs[grp[0]][0] not in col_names)
    res = dict(value=ab * steps / t["wall_s"] / 1e9, ms_per_step=t["wall_s"] * 1e3 / steps,
               us_per_decoder_layer=t["event_ms"] * 1e3 / steps / n_layers,
               decoder_layers=n_layers, projections=[p[0] for p in projs], hipgraph=t["captured"],
               alg_bytes_per_step=ab, regions_ms_per_step=t["regions_ms_per_step"],
               launches_per_decoder_layer=per_layer,
               all_reduce=((f"q/k/v/gate/up column-parallel (no collective), o / down row-parallel: {n_reduce} RCCL "
                            "all-reduces of fp32 partial outputs per decoder layer") if pair else
                           (f"fp32 partial outputs, {per_layer} RCCL all-reduces per decoder layer "
                            "(q+k+v, o, gate+up, down)") if fuse_siblings else
                           "fp32 partial outputs, one RCCL all-reduce per projection") if world > 1 else "none (1 rank)")
    if check and rank == 0:
        from oracle import c_oracle as co
        if co.available():
            errs = {}
            for name, xf, yv, rows in checks:   # the (reduced) output of one projection of every shape against the C oracle
                L = layer_spec(full0[name])
                xb = xf.detach().cpu().contiguous().view(torch.int16).numpy().view(np.uint16)
                want = co.forward(L, xb)
                if rows is not None:            # column-parallel: rank 0's slice of the output rows
                    want = np.ascontiguousarray(want.reshape(-1)[rows[0]:rows[1]])
                errs[name] = rel_err_bits(yv.contiguous(), want)
            res["parity_rel_err_vs_cpu_oracle"] = max(errs.values())
            res["parity_per_projection"] = errs
            assert res["parity_rel_err_vs_cpu_oracle"] <= 1e-3, res
    return res


COMPACT_LIMIT = 4096   # bytes: the driver keeps an 8 KB stdout tail; round 5's 21.5 KB line did not parse (BENCH_r05.parsed = null)


def _short_sample(s, n=160):
    s = str(s)
    return s if len(s) <= n else s[:n - 3] + "..."


def compact_line(out):
    """The ONE stdout line: the contract's keys + `roofline` + `cpu_baseline`, nothing verbose.  Everything else
    (`extras`, soak, clocks, notes) goes to gpurun_out/bench_full.json and stderr (write_full).  Pure function of the
    full result dict, tested on CPU (tests/test_bench_line_cpu.py): < COMPACT_LIMIT bytes, json round trip."""
    cfg = out.get("config") or {}
    rf = out.get("roofline") or {}
    c = {k: out.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                 "scaling", "vs_baseline", "dtype", "data")}
    arith = cfg.get("arithmetic") or ""
    c["config"] = {"workload": _short_sample(cfg.get("workload", ""), 260)}
    for k in ("hidden", "ring", "mode", "decoder_layers", "launches_per_step", "us_per_layer", "kernel", "hipgraph"):
        if cfg.get(k) is not None:
            c["config"][k] = cfg[k]
    c["config"]["arithmetic"] = next((m for m in ("reference", "selective", "folded") if arith.startswith(m)), arith[:40])
    c["config"]["parallelism"] = _short_sample(cfg.get("parallelism", ""), 120)
    r = {k: rf.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic")}
    for k in ("traffic_source", "traffic_from_committed_profile", "bytes_per_launch", "us_per_launch", "ms_per_step_one_step_per_graph",
              "power_w_during_timed_regions", "sclk_mhz_during_timed_regions"):
        if rf.get(k) is not None:
            r[k] = rf[k]
    mp = {}
    for name, e in (rf.get("module_path") or {}).items():
        row = {}
        for k_src, k_dst in (("us_per_layer", "us"), ("frac", "frac"), ("tokens_per_s", "tok_s"), ("vqlinear_us_per_token", "vq_us_tok")):
            if e.get(k_src) is not None:
                row[k_dst] = round(float(e[k_src]), 4)
        if e.get("kernel"):
            row["kernel"] = str(e["kernel"])[:40]
        mp[name] = row
    if mp:
        r["module_path"] = mp
    oi = rf.get("opt_in_arithmetic") or {}
    if oi:
        r["opt_in_arithmetic"] = {k: {kk: round(float(vv), 4) for kk, vv in v.items()} for k, v in oi.items()}
    bh = rf.get("bf16_headline_workload") or {}
    if bh:
        r["bf16_headline_workload"] = {k: (round(float(v), 4) if isinstance(v, (int, float)) else v) for k, v in bh.items()}
    c["roofline"] = r
    cb = out.get("cpu_baseline")
    if cb:
        c["cpu_baseline"] = {"value": cb.get("value"), "unit": cb.get("unit"), "cores": cb.get("cores"), "kind": cb.get("kind"),
                             "sample": _short_sample(cb.get("sample", ""), 200)}
    for k in ("parity_rel_err_vs_cpu_oracle", "process", "full"):
        if out.get(k) is not None:
            c[k] = out[k]
    tp = out.get("tp_row")
    if tp:
        c["tp_row"] = {k: tp.get(k) for k in ("us_per_decoder_layer", "allreduce_per_decoder_layer", "parity_rel_err_vs_cpu_oracle") if tp.get(k) is not None}
    ws = out.get("weak_scaling")
    if ws:
        c["weak_scaling"] = {k: ws.get(k) for k in ("value", "unit", "us_per_launch", "scaling")}
    line = json.dumps(c, separators=(",", ":"))
    if len(line) >= COMPACT_LIMIT:     # never let verbosity cost the record: drop the optional parts, in this order
        for victim in (("roofline", "opt_in_arithmetic"), ("roofline", "module_path"), ("config", "workload"), ("cpu_baseline", "sample")):
            obj = c.get(victim[0]) or {}
            if victim[1] in obj:
                obj[victim[1]] = None if victim[1] in ("module_path", "opt_in_arithmetic") else _short_sample(obj[victim[1]], 60)
            line = json.dumps(c, separators=(",", ":"))
            if len(line) < COMPACT_LIMIT:
                break
    return line


def write_full(out, path=None):
    """The whole result (extras, soak, clocks, notes): gpurun_out/bench_full.json (merged back by gpurun) + stderr."""
    path = path or os.path.join(ROOT, "gpurun_out", "bench_full.json")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            json.dump(out, f, indent=1)
        rel = os.path.relpath(path, ROOT)
    except OSError:
        rel = None
    print("[bench] full result: " + json.dumps(out), file=sys.stderr, flush=True)
    return rel


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--regions", type=int, default=5, help="timed regions of K steps; the median is reported")
    ap.add_argument("--hidden", type=int, default=8192)
    ap.add_argument("--ring", type=int, default=0)
    ap.add_argument("--mode", choices=["auto", "chain", "chain_dep", "single", "grouped", "tp", "tp_row", "tp_pair", "rings"], default="auto")
    ap.add_argument("--chain", type=int, default=32, help="chain: layers per launch (<= 32)")
    ap.add_argument("--group", type=int, default=4)
    ap.add_argument("--tp-layers", type=int, default=20,
                    help="tp_row: Llama-3-70B decoder layers in the ring (20 = 4.3 GB of packed indices)")
    ap.add_argument("--folded", "--fast-math", dest="folded", action="store_true",
                    help="the opt-in folded fp32 form sum (c+r)*f16(s*x) + sum b*x (vptq_amd.set_arithmetic('folded')) "
                         "instead of the default: every weight rebuilt with the reference CPU path's three 16-bit "
                         "roundings (VPTQ_GEMV_EXACT, bit-identical weights)")
    ap.add_argument("--exact", action="store_true", help="accepted, no effect: the default since round 5")
    ap.add_argument("--arithmetic", choices=["reference", "selective", "folded"], default=None,
                    help="the headline's arithmetic (default: reference = the product default; selective / folded = the opt-in forms, "
                         "vptq_amd.set_arithmetic)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--extras", action="store_true",
                    help="measure every other route too (folded opt-in, grouped, tokens, k8192, wide layers, tp_row on one GPU, prefill, the "
                         "Llama-3-8B-shaped decode loop: ~60 s more).  Default: only the module-path rows of the compact line "
                         "(one launch per layer at 8192^2 and 4096^2, the large-codebook formats' default route)")
    ap.add_argument("--prefetch", action="store_true")
    ap.add_argument("--steps-per-graph", type=int, default=0,
                    help="steps captured per hipGraph (0 = as many as divide K, up to 10; 1 = the rounds 1 - 3 methodology: "
                         "one graph launch per step)")
    a = ap.parse_args()
    if a.steps_per_graph > 0:
        Timer.MAX_STEPS_PER_GRAPH = a.steps_per_graph

    # stdout carries exactly ONE line, the JSON: everything else written to file descriptor 1 - RCCL prints
    # its version banner there, hipBLASLt / MIOpen may warn - goes to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        # the full result to gpurun_out/bench_full.json + stderr, the compact line (LAST thing on stdout) to the driver
        obj["full"] = write_full(obj)
        sys.stdout.flush()
        sys.stderr.flush()
        os.write(json_fd, (compact_line(obj) + "\n").encode())

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus and world == 1 and a.gpus > 1:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    # VPTQ_BENCH_SAME_GPU=1 (test knob, tools/gpu_bench_2ranks.sh): every rank on GPU 0 with gloo collectives, so
    # that the N > 1 code path (shards by rank, collectives inside the timed region, parity of the reduced
    # result, max over ranks) runs end to end on a one-GPU box.  The numbers of such a run mean nothing.
    same_gpu = os.environ.get("VPTQ_BENCH_SAME_GPU") == "1"
    if same_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # one process per GPU: LOCAL_RANK is the device (every launch, allocation and the process group's device below)
    assert torch.cuda.current_device() == local_rank and dev.index == local_rank, (torch.cuda.current_device(), local_rank)
    dist = None
    # VPTQ_BENCH_FORCE_DIST=1 (test knob, tests/test_bench_dist_gpu.py): initialise the process group at world size 1 too, so
    # that a one-GPU box runs the N > 1 code path through the REAL backend - RCCL init, the all-reduce inside the step, its
    # hipGraph capture (or the eager fallback) - before the first multi-GPU run ever happens
    force_dist = os.environ.get("VPTQ_BENCH_FORCE_DIST") == "1"
    if world > 1 or force_dist:
        import torch.distributed as dist
        if same_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    mode = a.mode
    if mode == "auto":
        mode = "chain" if (world == 1 and not force_dist) else "tp_row"
    if mode == "rings":
        mode = "single"

    from vptq_amd import _backend as B
    lib = B.lib()
    H = a.hidden
    if a.arithmetic is None:
        a.arithmetic = "folded" if a.folded else "reference"
    a.folded = a.arithmetic == "folded"
    a.exact = a.arithmetic == "reference"
    flags = {"reference": B.GEMV_EXACT, "selective": B.GEMV_SELECTIVE, "folded": 0}[a.arithmetic]
    B.set_arithmetic(a.arithmetic)   # (module routes inside the extras follow the same mode)
    timer = Timer(dev, dist, allow_eager=mode in ("tp", "tp_row", "tp_pair"))
    arithmetic = ("reference roundings per weight (VPTQ_GEMV_EXACT, the product default since round 5): w = f16(f16(f16(c+r)*s)+b) "
                  "as the reference CPU path rounds it, fp32 accumulate, one rounding of y - bit-identical weights, >= 99 % of "
                  "the outputs bit-identical" if a.exact else
                  "selective (opt-in, --arithmetic selective / VPTQ_ARITHMETIC=selective, round 6): the folded form with the reference's roundings "
                  "on the 128-column blocks an activation dominates (|f16(s x)| >= 6 rms); 2 of 12 300 checkpoint-like layers above the bar at "
                  "1.00e-3 / 1.09e-3 through the chain route (folded: 30 of 4100), profiles/r06/count_chain_*.txt" if a.arithmetic == "selective" else
                  "folded fp32 (opt-in, --folded / VPTQ_ARITHMETIC=folded): sum (c+r)*f16(s*x) + sum b*x - inside the 1e-3 "
                  "max-normalised parity bar for dense activations (measured 5-6e-4), not bit-equivalent, above the bar on 3 of 1000 "
                  "checkpoint-like layers with massive-channel / sparse activations (profiles/r05/gate_count_*)")

    if mode in ("tp_row", "tp_pair"):
        r = bench_tp_row(lib, B, dev, timer, rank, world, a.tp_layers, flags, a.steps, a.warmup, a.regions,
                         pair=mode == "tp_pair")
        out = {
            "metric": "decode GEMV effective GB/s (VQuantLinear 2-bit, batch 1)",
            "value": r["value"], "unit": "GB/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": r["ms_per_step"], "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": f"Llama-3-70B shaped decoder layers x {a.tp_layers} (q/o 8192^2, k/v 1024x8192, "
                                   "gate/up 28672x8192, down 8192x28672; v=8 k=256+256, 2-bit), batch=1 seq=1 fp16, "
                                   f"every projection row-parallel over {world} rank(s): fused GEMV with fp32 partial "
                                   "output -> RCCL all-reduce -> one rounding; 1 step = 1 token through all layers' "
                                   "projections", "mode": mode, "decoder_layers": a.tp_layers,
                       "hipgraph": r["hipgraph"], "arithmetic": arithmetic,
                       "parallelism": f"tp{world} row-parallel (input columns), RCCL all-reduce per projection",
                       "strong_scaling_baseline": "python bench.py --mode tp_row --gpus 1 (also: extras.tp_row_n1 of "
                                                  "the N = 1 line)"},
            "regions_ms_per_step": r["regions_ms_per_step"],
            "roofline": {"bound": "hbm", "achieved": r["value"] / world, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": r["value"] / world / HBM_PEAK_GBPS, "traffic": None,
                         "note": "per GPU: whole-job algorithmic bytes / N / time, collectives included"},
            "tp_row": {k: v for k, v in r.items() if k != "regions_ms_per_step"},
            "process": {"rank": rank, "local_rank": local_rank, "world": world, "device": torch.cuda.current_device(),
                        "collective_backend": None if dist is None else dist.get_backend()},
        }
        if world > 1 and not a.no_extras and mode == "tp_row":
            # the Megatron pairing (2 all-reduces per decoder layer instead of 4) on the same layers
            pr = bench_tp_row(lib, B, dev, Timer(dev, dist, allow_eager=True), rank, world, a.tp_layers, flags,
                              max(5, a.steps // 4), 5, 3, pair=True)
            out["tp_pair"] = {k: v for k, v in pr.items() if k != "regions_ms_per_step"}
        if world > 1 and not a.no_extras:
            w, *_ = bench_ring(lib, B, dev, Timer(dev, dist), H, "single", flags, max(5, a.steps // 4), 5, 3,
                               rank=rank, world=world)
            out["weak_scaling"] = {"what": f"{world} x independent rings of 8192^2 layers, no collective",
                                   "value": w["value"], "unit": "GB/s", "us_per_launch": w["us_per_launch"],
                                   "scaling": "weak"}
        if rank == 0:
            emit(out)
        if dist is not None:
            dist.destroy_process_group()
        return

    tp_split = "out" if mode == "tp" else None
    if mode == "tp" and H // 8 % world:
        raise SystemExit("tp mode needs the vector-row count divisible by the world size")
    with SclkSampler(local_rank) as sclk:
        r, layers, x, ys, keeps = bench_ring(lib, B, dev, timer, H, mode if mode in ("grouped", "chain", "chain_dep") else "single",
                                             flags, a.steps, a.warmup, a.regions, rank=rank, world=world, group=a.group,
                                             ring=a.ring, tp_split=tp_split, prefetch=a.prefetch, chain=a.chain)
    soak = None
    if rank == 0 and world == 1 and os.environ.get("VPTQ_BENCH_SOAK", "1") != "0":
        # what the package draws and where the shader clock settles while this workload runs back to back for 1.5 s
        with SclkSampler(local_rank) as sk:
            us_replay = timer.soak(1.5)
        if sk.power_summary() is not None or sk.summary() is not None:
            soak = {"what": "the timed workload replayed back to back for 1.5 s (after the timed regions), hwmon of this GPU sampled every 20 ms",
                    "us_per_step": us_replay, "power": sk.power_summary(), "sclk": sk.summary(), "mclk_mhz": sk.mclk_mhz()}
    chain_mode = mode in ("chain", "chain_dep")
    mode_name = (f"chain{min(a.chain, 32, r['ring'])}" if mode == "chain" else mode)
    out = {
        "metric": "decode GEMV effective GB/s (VQuantLinear 2-bit, batch 1)",
        "value": r["value"], "unit": "GB/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": r["ms_per_step"], "higher_is_better": True,
        "scaling": "strong" if tp_split else "weak",
        "vs_baseline": None, "dtype": "f16", "data": "synthetic",
        "config": {"workload": f"VQuantLinear {H}x{H} v=8 k=256+256 (2-bit) batch=1 seq=1 fp16, "
                               f"ring of {r['ring']} distinct layers per GPU ({r['idx_mib']} MiB of "
                               f"packed indices), 1 step = 1 pass over the ring"
                               + (f" = {r['launches_per_step']} persistent launch(es) walking "
                                  f"{'dependent' if mode == 'chain_dep' else 'independent'} layers one after the other"
                                  if chain_mode else ""),
                   "hidden": H, "ring": r["ring"], "mode": mode_name,
                   "mode_note": ("throughput mode: the ring's layers are INDEPENDENT (each its own x and y), which is what lets one persistent "
                                 "launch walk 32 of them; the layers of a decoder depend on each other and run at the rate of "
                                 "extras.single_launch_per_layer (the reference's operator granularity), sibling projections at "
                                 "extras.grouped_x4") if mode == "chain" else None,
                   "launches_per_step": r["launches_per_step"],
                   "us_per_layer": r["us_per_layer"],
                   "kernel": r["kernel"], "arithmetic": arithmetic,
                   "read_ahead_next_layer": bool(a.prefetch), "hipgraph": r["hipgraph"],
                   "timing": (f"median of {a.regions} regions of {a.steps} steps; a region = {a.steps // r['steps_per_graph']} replays of a "
                              f"hipGraph holding {r['steps_per_graph']} steps; before the {a.warmup} warm-up steps the workload is replayed "
                              f"for {r['conditioning_s']:g} s so that clocks and package power are in their sustained state (an idle "
                              "MI355X sits at 160 MHz; bursts of a few ms run faster than the sustained rate)"),
                   "parallelism": (f"tp{world}: output rows of every layer split over {world} ranks, "
                                   "RCCL all-gather per layer") if mode == "tp" else
                                  f"{world} x independent rings (no collective)"},
        "regions_ms_per_step": r["regions_ms_per_step"],
        "sclk_during_timed_regions": sclk.summary(),
        "power_during_timed_regions": sclk.power_summary(),
        "soak": soak,
        "roofline": {"bound": "hbm", "achieved": r["achieved"], "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": r["achieved"] / HBM_PEAK_GBPS, "traffic": None,
                     "bytes_per_launch": r["bytes_per_launch"], "us_per_launch": r["us_per_launch"],
                     "regions_ms_per_step": r["regions_ms_per_step"], "steps_per_graph": r["steps_per_graph"],
                     "ms_per_step_one_step_per_graph": r["ms_per_step_one_step_per_graph"],
                     "conditioning_s": r["conditioning_s"],
                     "power_w_during_timed_regions": (sclk.power_summary() or {}).get("median_w"),
                     "sclk_mhz_during_timed_regions": (sclk.summary() or {}).get("median_mhz"),
                     "soak": None if soak is None else {"us_per_step": soak["us_per_step"],
                                                        "power_w": (soak["power"] or {}).get("median_w"),
                                                        "sclk_mhz": (soak["sclk"] or {}).get("median_mhz"),
                                                        "mclk_mhz": soak.get("mclk_mhz"),
                                                        "reading": "at the 1400 W cap with the shader clock pulled down = the energy-bound state of DESIGN 4.9; "
                                                                   "well under the cap at a high shader clock = this box's memory system is the limit (boxes of the pool: "
                                                                   "143 us per step at 1400 W / 1.66 GHz on most, 237 us at 1020 W / 2.1 GHz on one)"},
                     "note": ("one launch = the whole ring (32 layers): bytes_per_launch = 32 x the algorithmic bytes of a layer (SURVEY 8d), "
                              "us_per_launch = HIP-event time over the (median) timed region / launches.  Reference roundings: 12 packed-f16 "
                              "instructions per index rebuild the 8 weights as the reference rounds them (add, multiply by the column's scale, add "
                              "its bias: three roundings) + 2 MFMA 4x4x4 for the products - the launch is bound by VECTOR ISSUE, not by HBM "
                              "(DESIGN.md 4.9: the same loop with the folded arithmetic - 4 v_perm + 4 MFMA per index - runs at the package "
                              "power limit, extras.folded_chain)") if (chain_mode and a.exact) else
                             ("one launch = the whole ring (32 layers): bytes_per_launch = 32 x the algorithmic bytes of a "
                              "layer (SURVEY 8d), us_per_launch = HIP-event time over the (median) timed region / launches; "
                              "what bounds it: the 1400 W package power limit - this kernel draws the cap and the shader clock "
                              "settles at ~1.65 of 2.4 GHz (`soak` in this object).  Energy roofline (profiles/r04/"
                              "ubench_energy_table.txt, DESIGN.md 4.9): 351 W with resident idle waves, 98.7 pJ per HBM byte = 1.66 mJ "
                              "per layer, 19-20 nJ per index-wave for gathers + multiply-accumulate in EVERY formulation measured "
                              "(4 MFMA 4x4x4; f16(c+r) + 2 MFMA; packed-f16 VALU; fp32 VALU) = 2.6 mJ per layer: (1.66 + 2.6) mJ / "
                              "(1400 - 351) W = 4.1 us per layer = 0.51 of 8 TB/s is the ceiling of this format at one token under this "
                              "cap; 0.70 would need <= 0.9 mJ per layer outside the memory system, the LDS gathers alone are 1.1") if chain_mode else
                             ("us_per_launch = HIP-event time over the (median) timed region / launches, i.e. "
                              "INCLUDING the kernel boundary (an empty kernel in the same graph: 1.8 us per "
                              "launch, profiles/r02/ubench_stream_8192.txt); rocprofv3's kernel-only duration is "
                              "shorter by the idle part of that boundary.  The kernel is bound by instruction "
                              "issue / LDS gathers and per-launch latency, not by HBM: compute_floor, DESIGN.md 4"),
                     "compute_floor": None if chain_mode else compute_floor(r["kernel"], H)},
    }
    # the newest round's PMC summary of this configuration
    rounds = sorted(d for d in os.listdir(os.path.join(ROOT, "profiles")) if d[:1] == "r" and d[1:].isdigit())
    for rd in reversed(rounds):
        cand = os.path.join(ROOT, "profiles", rd, f"bench_h{H}_{'chain' if chain_mode else mode}"
                            f"{'_exact' if a.exact else '_selective' if a.arithmetic == 'selective' else ''}_pmc_summary.json")
        if os.path.exists(cand) and not a.prefetch:
            # HBM bytes per launch from a separate rocprofv3 --pmc run of this same command
            # (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE); see the file's _note
            out["roofline"]["traffic"] = json.load(open(cand)).get("hbm_bytes_corrected")
            out["roofline"]["traffic_source"] = os.path.relpath(cand, ROOT)
            out["roofline"]["traffic_from_committed_profile"] = True   # NOT measured in this run: a PMC pass cannot share a process with the timing
            break
    if rank == 0 and world == 1 and not a.no_cpu_baseline and mode in ("single", "grouped", "chain", "chain_dep"):
        torch.cuda.synchronize()
        # parity on the LAST layer of the ring (of the chain), against the C oracle on the same bits
        x_last = ys[-2] if mode == "chain_dep" else x
        base, rel = cpu_baseline(layers[-1], x_last, ys[-1], H)
        out.update(base)
        assert rel <= 1e-3, f"GPU result differs from the CPU oracle: {rel}"
        # ... and on the first layer and two inside the ring (C oracle on the same bits; the torch restatement and the
        # CPU timing above are the last layer's)
        checked = {len(layers) - 1: rel}
        if mode != "chain_dep":
            from oracle import c_oracle as co
            if co.available():
                xb = x.detach().cpu().contiguous().view(torch.int16).numpy().view(np.uint16)
                for i in sorted({0, len(layers) // 3, 2 * len(layers) // 3} - {len(layers) - 1}):
                    checked[i] = rel_err_bits(ys[i], co.forward(layer_spec(layers[i]), xb))
                    assert checked[i] <= 1e-3, f"layer {i}: GPU result differs from the CPU oracle: {checked[i]}"
        out["parity_checked_on"] = {"layers_of_the_ring": sorted(checked), "rel_err_vs_cpu_oracle": [checked[i] for i in sorted(checked)],
                                    "bar": 1e-3, "every_layer": "tests/test_chain_gpu.py::test_chain_of_32_distinct_8192_layers_every_output"}
    if world == 1 and not a.no_extras and mode in ("single", "chain") and a.exact:
        del layers, ys, keeps
        torch.cuda.empty_cache()
        ex = {}
        st, wu, rg = max(10, a.steps // 4), 5, 3
        EX = B.GEMV_EXACT

        def short(res, nbytes=None):
            return {"GBps": res["achieved"], "frac_of_8TBps": res["achieved"] / HBM_PEAK_GBPS,
                    "us_per_launch": res["us_per_launch"], "kernel": res["kernel"]}
        # (name, arguments, what) - the product default (reference roundings) first, the opt-in folded form of the same
        # workloads as folded_*
        table = (
            ("single_launch_per_layer", dict(H=H, mode="single", flags=EX),
             "the same ring, one vptq_quant_gemv launch per layer - what VQuantLinear.forward issues (the reference's operator granularity)"),
            ("h4096_chain", dict(H=4096, mode="chain", flags=EX), "VQuantLinear 4096x4096 (BASELINE configs[0]/[1] shape), ring of 128 layers as 4 chain launches"),
            ("h4096", dict(H=4096, mode="single", flags=EX), "VQuantLinear 4096x4096 (BASELINE configs[0]/[1] shape), single launch per layer"),
            ("selective_single_launch_per_layer", dict(H=H, mode="single", flags=B.GEMV_SELECTIVE),
             "OPT-IN selective arithmetic, one launch per layer (what VQuantLinear.forward issues after vptq_amd.set_arithmetic('selective')): "
             "gemv_k256m_kernel<selective> - hot blocks found, zeroed and corrected inside the launch"),
            ("bf16_single_launch_per_layer", dict(H=H, mode="single", flags=EX, dtype=torch.bfloat16),
             "bf16 layers (the dtype of most published checkpoints), one launch per layer, the default arithmetic: the reference's three roundings "
             "on the matrix pipe (round 6: one-hot MFMA operands widen for free; 21.7 us as widened VALU arithmetic before)"),
            ("bf16_chain", dict(H=H, mode="chain", flags=EX, dtype=torch.bfloat16),
             "bf16 layers, the headline workload (32 layers per persistent launch) in the default arithmetic: the same matrix-pipe roundings inside gemv_k256c"),
            ("selective_bf16_single_launch_per_layer", dict(H=H, mode="single", flags=B.GEMV_SELECTIVE, dtype=torch.bfloat16),
             "bf16 layers, OPT-IN selective arithmetic: the dtype-agnostic folded MFMA loop + widened corrections on the hot blocks only"),
            ("selective_chain", dict(H=H, mode="chain", flags=B.GEMV_SELECTIVE),
             "OPT-IN selective arithmetic (vptq_amd.set_arithmetic('selective'), round 6), the headline workload: the folded form with the "
             "reference's roundings on the 128-column blocks an activation dominates (|f16(s x)| >= 6 rms) - 2 of 12 300 checkpoint-like layers "
             "above the bar at 1.00e-3 / 1.09e-3 against 30 of 4100 for the folded form, profiles/r06/count_chain_*.txt"),
            ("grouped_x4", dict(H=H, mode="grouped", flags=EX), "4 independent layers per launch (vptq_quant_gemv_grouped), us per LAYER = us_per_launch / 4"),
            ("folded_chain", dict(H=H, mode="chain", flags=0),
             "OPT-IN folded arithmetic (vptq_amd.set_arithmetic('folded')), the headline workload: 32 layers per persistent launch (the headline of rounds 3-4)"),
            ("folded_single_launch_per_layer", dict(H=H, mode="single", flags=0), "opt-in folded arithmetic, one launch per layer"),
            ("folded_h4096", dict(H=4096, mode="single", flags=0), "opt-in folded arithmetic, 4096x4096, one launch per layer"),
            ("folded_h4096_chain", dict(H=4096, mode="chain", flags=0), "opt-in folded arithmetic, 4096x4096, chains of 32"),
            ("chain_dependent", dict(H=H, mode="chain_dep", flags=0),
             "the same ring as ONE dependent chain (x of layer i + 1 = y of layer i; folded arithmetic only): device-scope hand-over per layer "
             "inside the launch; us_per_launch is the whole chain"),
            ("tokens16", dict(H=H, mode="single", flags=EX, tokens=16), "16 tokens per launch (batched decode, reference roundings), bytes incl. 16 x and y rows"),
            ("folded_tokens16", dict(H=H, mode="single", flags=0, tokens=16), "16 tokens in ONE pass over the indices (gemm_k256t, folded arithmetic only)"),
            ("folded_tokens16_bf16", dict(H=H, mode="single", flags=0, tokens=16, dtype=torch.bfloat16),
             "16 bf16 tokens in one pass over the indices (gemm_k256t: transposing gathers -> 16x16x32 MFMA, tokens = M; + its pre-pass)"),
            ("k8192_r256", dict(H=H, mode="single", flags=EX, k=8192, kr=256), "k = 8192 + 256 (T = 21 bits), LDS-resident codebooks, reference roundings"),
            ("folded_k8192_r256", dict(H=H, mode="single", flags=0, k=8192, kr=256), "k = 8192 + 256, opt-in folded arithmetic (MFMA accumulate)"),
        )
        core = ("single_launch_per_layer", "h4096", "selective_chain", "selective_single_launch_per_layer",
                "bf16_single_launch_per_layer", "bf16_chain", "selective_bf16_single_launch_per_layer")   # the compact line's roofline.module_path / opt_in rows
        for key, kw, what in table:
            if not a.extras and key not in core:
                continue
            kw = dict(kw)
            try:   # (the headline line must not depend on an extra)
                rr, *_ = bench_ring(lib, B, dev, Timer(dev), kw.pop("H"), kw.pop("mode"), kw.pop("flags"), st, wu, rg, **kw)
                ex[key] = short(rr)
                ex[key]["us_per_layer"] = rr["us_per_layer"]
            except Exception as e:
                ex[key] = {"error": f"{type(e).__name__}: {e}"[:300]}
            ex[key]["what"] = what
            torch.cuda.empty_cache()
        for key, kw in (("k65536_r256", {}), ("k65536_r65536", dict(kr=65536)),     # ... the "4 bits" format of every published family
                        ("v16_k65536_r65536", dict(kr=65536, v=16))):              # "2 bits" of most families
            try:   # (the headline line must not depend on an extra)
                ex[key] = k65536_extra(lib, B, dev, H, st, wu, rg, light=not a.extras, R=8 if a.extras else 4, **kw)
            except Exception as e:
                ex[key] = {"error": f"{type(e).__name__}: {e}"[:300]}
                torch.cuda.empty_cache()
        try:   # (the headline line must not depend on an extra)
            if a.extras:
                ex["k65536_r256_28672x8192"] = wide_layer_extra(lib, B, dev, st, wu, rg)
        except Exception as e:
            ex["k65536_r256_28672x8192"] = {"error": f"{type(e).__name__}: {e}"[:300]}
            torch.cuda.empty_cache()
        try:
            if not a.extras:
                raise StopIteration
            tr = bench_tp_row(lib, B, dev, Timer(dev), 0, 1, a.tp_layers, EX, st, wu, rg)
            ex["tp_row_n1"] = {"what": f"Llama-3-70B shaped decoder layers (x{a.tp_layers}, the ring of --gpus N) on ONE GPU "
                                       "through the row-parallel code path (world size 1): the strong-scaling baseline of --gpus N",
                               "GBps": tr["value"], "us_per_decoder_layer": tr["us_per_decoder_layer"],
                               "parity_rel_err_vs_cpu_oracle": tr.get("parity_rel_err_vs_cpu_oracle"),
                               "parity_per_projection": tr.get("parity_per_projection")}
        except StopIteration:
            pass
        except Exception as e:
            ex["tp_row_n1"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        try:
            if a.extras:
                ex["prefill"] = prefill_extra(dev)
        except Exception as e:   # the headline line must not depend on it
            ex["prefill"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        # (the opt-in folded arithmetic of the same loop: VPTQ_BENCH_FOLDED_DECODE=1 - 25 s more; profiles/r05/bench_h8192_chain.json has it)
        for key, mode_ in ((("llama3_8b_decode", "reference"),) if a.extras else ()) + ((("folded_llama3_8b_decode", "folded"),) if os.environ.get("VPTQ_BENCH_FOLDED_DECODE") == "1" else ()):
            try:
                ex[key] = model_decode_extra(arithmetic=mode_)
            except Exception as e:
                ex[key] = {"error": f"{type(e).__name__}: {e}"[:300]}
        out["extras"] = ex
        # what the drop-in module delivers (VQuantLinear.forward -> ops.quant_gemm -> ONE vptq_quant_gemv launch per layer, the
        # reference's operator granularity, vptq/ops/quant_gemm.py:213-228), beside the throughput-mode headline: the driver
        # keeps the `roofline` object whole, `extras` only in part
        mp = {}
        for key, name in (("single_launch_per_layer", f"h{H}"), ("h4096", "h4096"), ("selective_single_launch_per_layer", f"h{H}_selective_opt_in"),
                          ("bf16_single_launch_per_layer", f"h{H}_bf16"), ("selective_bf16_single_launch_per_layer", f"h{H}_bf16_selective_opt_in"),
                          ("folded_single_launch_per_layer", f"h{H}_folded_opt_in"), ("folded_h4096", "h4096_folded_opt_in")):
            e = ex.get(key) or {}
            if "us_per_launch" in e:
                mp[name] = {"us_per_layer": e["us_per_launch"], "GBps": e["GBps"], "frac": e["frac_of_8TBps"], "kernel": e["kernel"]}
        for key in ("k65536_r256", "k65536_r65536", "v16_k65536_r65536"):
            e = ex.get(key) or {}
            sl = e.get("sliced_layout") or {}
            df = e.get("exact_sliced_layout") or e.get("default") or {}
            if "us_per_layer" in df:     # the product default: the exact sliced kernel where it serves the format, else centroid gathers through the caches
                mp[key] = {"us_per_layer": df["us_per_layer"], "GBps_of_packed_bytes": df.get("GBps"), "frac": df.get("frac_of_8TBps"),
                           "kernel": df.get("kernel")}
            ss = e.get("selective_sliced_layout") or {}
            if "us_per_layer" in ss:     # ... the opt-in selective arithmetic (round 6)
                mp[key + "_selective_opt_in"] = {"us_per_layer": ss["us_per_layer"], "GBps_of_packed_bytes": ss.get("GBps"),
                                                 "frac": ss.get("frac_of_8TBps"), "kernel": "gemv_hot + gemv_sliced"}
            if "us_per_layer" in sl:     # ... and the opt-in folded arithmetic over the load-time derived sliced layouts
                mp[key + "_folded_opt_in"] = {"us_per_layer": sl["us_per_layer"], "GBps_of_packed_bytes": sl.get("GBps"),
                                              "frac": sl.get("frac_of_8TBps"), "kernel": sl.get("kernel", "gemv_sliced_kernel")}
        dec = ex.get("llama3_8b_decode") or {}
        if "vqlinear_us_per_token" in dec:
            mp["llama3_8b_decode_step"] = {k_: dec.get(k_) for k_ in ("tokens_per_s", "vqlinear_us_per_token", "vqlinear_GBps")}
        dec = ex.get("folded_llama3_8b_decode") or {}
        if "vqlinear_us_per_token" in dec:
            mp["llama3_8b_decode_step_folded_opt_in"] = {k_: dec.get(k_) for k_ in ("tokens_per_s", "vqlinear_us_per_token", "vqlinear_GBps")}
        out["roofline"]["module_path"] = mp
        # the opt-in arithmetics on the headline workload (never `value`): selective (round 6) and, with --extras, folded
        oi = {}
        for key, name in (("selective_chain", "selective"), ("folded_chain", "folded")):
            e = ex.get(key) or {}
            if "us_per_layer" in e:
                oi[name] = {"GBps": e["GBps"], "frac": e["frac_of_8TBps"], "us_per_layer": e["us_per_layer"]}
        out["roofline"]["opt_in_arithmetic"] = oi
        e = ex.get("bf16_chain") or {}
        if "us_per_layer" in e:   # the headline workload on bf16 layers, default arithmetic (never `value`: BASELINE's metric is quoted on fp16)
            out["roofline"]["bf16_headline_workload"] = {"GBps": e["GBps"], "frac": e["frac_of_8TBps"], "us_per_layer": e["us_per_layer"], "kernel": e["kernel"]}
    if rank == 0:
        emit(out)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
