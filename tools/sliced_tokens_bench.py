#!/usr/bin/env python3
"""2-4 tokens of the large-codebook formats: the gather kernels (one launch for all tokens) against one sliced launch PER TOKEN
(VQuantLinear._gemv_cached, VPTQ_SLICED_TOKENS) and against ONE sliced launch for all tokens (gemv_sliced_tok.hip), ring of distinct
layers in a hipGraph; us per layer.
    python tools/sliced_tokens_bench.py --v 8 --kr 65536 --shapes "8192,8192;4096,4096" """
import argparse, json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ["VPTQ_TUNING"] = "1"          # (tuning knobs are read only with it)
os.environ["VPTQ_SLICED_TOKENS"] = "4,4"   # (the route under test is chosen per call below)
os.environ["VPTQ_SLICED_ONE_LAUNCH"] = "0"  # (the module's forward = one sliced launch PER token here; the one-launch kernel is called directly)
from microbench import time_graph  # noqa
from shape_bench import mk  # noqa

ap = argparse.ArgumentParser()
ap.add_argument("--shapes", default="8192,8192;4096,4096")
ap.add_argument("--ring", type=int, default=8)
ap.add_argument("--k", type=int, default=65536)
ap.add_argument("--kr", type=int, default=0)
ap.add_argument("--v", type=int, default=8)
ap.add_argument("--no-eight", action="store_true")
ap.add_argument("--siblings", default="", help="O1,O2,O3 with --shapes I,*: q / k / v as three launches against one grouped launch, 1 - 4 tokens")
ap.add_argument("--only-one-launch", action="store_true", help="time only the one-launch kernel (ablation builds: results are wrong)")
a = ap.parse_args()
dev = torch.device("cuda", 0); g = torch.Generator(device=dev).manual_seed(0)
if a.siblings:
    from vptq_amd.utils.sliced import SlicedGemv, SlicedGroupGemv
    I = int(a.shapes.split(';')[0].split(',')[0])
    outs = [int(o) for o in a.siblings.split(',')]
    groups = []
    for _ in range(a.ring):
        ms = [mk(I, O, dev, g, k=a.k, kr=a.kr, v=a.v) for O in outs]
        sls = [SlicedGemv(m) for m in ms]
        groups.append((ms, sls, SlicedGroupGemv(sls)))
    row = dict(I=I, outs=outs, v=a.v, k=a.k, kr=a.kr)
    for T in (1, 2, 3, 4, 6, 8):
        x = torch.randn(1, T, I, device=dev).half()
        if T == 1:
            f_alone = lambda: [[sl(x) for sl in sls] for _, sls, _ in groups]
            f_group = lambda: [sg(x) for _, _, sg in groups]
        else:
            if not all(sg.tokens_supported(T) for _, _, sg in groups):
                continue
            f_alone = lambda: [[sl.forward_tokens(x) for sl in sls] for _, sls, _ in groups]
            f_group = lambda: [sg.forward_tokens(x) for _, _, sg in groups]
        f_alone(); f_group()
        row[f"t{T}"] = dict(three_launches_us=round(time_graph(f_alone, 10) / a.ring, 2), one_launch_us=round(time_graph(f_group, 10) / a.ring, 2))
    print(json.dumps(row), flush=True)
    sys.exit(0)
for I, O in [tuple(int(v) for v in p.split(',')) for p in a.shapes.split(';')]:
    layers = [mk(I, O, dev, g, k=a.k, kr=a.kr, v=a.v) for _ in range(a.ring)]
    row = dict(I=I, O=O, v=a.v, k=a.k, kr=a.kr)
    for T in ((1, 2, 3, 4, 6, 8) if a.only_one_launch else (1, 2, 3, 4)):
        x = torch.randn(1, T, I, device=dev).half()
        for m in layers:
            m.enable_sliced_layout(True)
        if a.only_one_launch:
            if T >= 2 and all(m._sliced_gemv().tokens_supported(T) for m in layers):
                [m._sliced_gemv().forward_tokens(x) for m in layers]
                row[f"t{T}"] = round(time_graph(lambda: [m._sliced_gemv().forward_tokens(x) for m in layers], 10) / a.ring, 2)
            continue
        ys = [m(x) for m in layers]           # builds the layouts
        us_s = time_graph(lambda: [m(x) for m in layers], 10) / a.ring
        us_t, err_t = None, None
        if T >= 2 and all(m._sliced_gemv() is not None and m._sliced_gemv().tokens_supported(T) for m in layers):
            # ONE launch for the T tokens (gemv_sliced_tok.hip: column phases)
            yt = [m._sliced_gemv().forward_tokens(x) for m in layers]
            us_t = time_graph(lambda: [m._sliced_gemv().forward_tokens(x) for m in layers], 10) / a.ring
        for m in layers:
            m.enable_sliced_layout(False)
        yg = layers[0](x)
        us_g = time_graph(lambda: [m(x) for m in layers], 10) / a.ring
        err = ((ys[0].float() - yg.float()).abs().max() / yg.float().abs().max()).item()
        row[f"t{T}"] = dict(gather_us=round(us_g, 2), sliced_per_token_us=round(us_s, 2), rel_diff=err)
        if us_t is not None:
            row[f"t{T}"]["sliced_one_launch_us"] = round(us_t, 2)
            row[f"t{T}"]["one_launch_rel_diff"] = ((yt[0].float() - yg.float()).abs().max() / yg.float().abs().max()).item()
    if not a.only_one_launch and not a.no_eight:
        # 5 - 8 tokens: the gather kernels (one launch for v = 8, two for v = 16) against TWO launches over the layouts (4 + the rest)
        for T in (6, 8):
            x = torch.randn(1, T, I, device=dev).half()
            for m in layers:
                m.enable_sliced_layout(True)
            sls = [m._sliced_gemv() for m in layers]
            us_t = None
            if all(sl is not None and sl.tokens_supported(4) and sl.tokens_supported(T - 4) for sl in sls):
                xa, xb = x[:, :4].contiguous(), x[:, 4:].contiguous()
                [(sl.forward_tokens(xa), sl.forward_tokens(xb)) for sl in sls]
                us_t = time_graph(lambda: [(sl.forward_tokens(xa), sl.forward_tokens(xb)) for sl in sls], 10) / a.ring
            for m in layers:
                m.enable_sliced_layout(False)
            layers[0](x)
            us_g = time_graph(lambda: [m(x) for m in layers], 10) / a.ring
            row[f"t{T}"] = dict(gather_us=round(us_g, 2), sliced_two_launches_us=None if us_t is None else round(us_t, 2))
    print(json.dumps(row), flush=True)
