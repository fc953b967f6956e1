#!/usr/bin/env python3
"""v8-k65536-0 layers, one token: the library's default route (gemv_gather_kernel: centroid gathers from L2)
against the GEMV over the load-time derived sliced layout (gemv_sliced.hip), ring of distinct layers in a hipGraph.
    python tools/sliced_bench.py --shapes "8192,8192;4096,4096" """
import argparse, json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from vptq_amd import _backend as B  # noqa
from vptq_amd.utils.sliced import SlicedGemv  # noqa
from _gpu_util import module_desc  # noqa
from microbench import time_graph  # noqa
from shape_bench import mk  # noqa

ap = argparse.ArgumentParser()
ap.add_argument("--shapes", default="8192,8192;4096,4096")
ap.add_argument("--ring", type=int, default=8)
ap.add_argument("--rpw", type=int, default=0)
ap.add_argument("--kr", type=int, default=0, help="residual centroids: 0 (v8-k65536-0) or 256 (v8-k65536-256)")
ap.add_argument("--k", type=int, default=65536, help="main centroids: 16384 / 32768 / 65536")
ap.add_argument("--v", type=int, default=8, help="vector length: 8 or 16 (v16-k65536-0 / -65536)")
ap.add_argument("--bf16", action="store_true")
ap.add_argument("--exact", action="store_true", help="the reference's roundings (VPTQ_GEMV_EXACT): gather kernel against the exact sliced kernel")
ap.add_argument("--out", default="")
a = ap.parse_args()
dev = torch.device("cuda", 0); g = torch.Generator(device=dev).manual_seed(0); lib = B.lib()
dt = torch.bfloat16 if a.bf16 else torch.float16
res = []
for I, O in [tuple(int(v) for v in p.split(',')) for p in a.shapes.split(';')]:
    R = a.ring
    layers = [mk(I, O, dev, g, k=a.k, kr=a.kr, v=a.v) for _ in range(R)]
    if a.bf16:
        layers = [m.to(torch.bfloat16) for m in layers]
    descs = [module_desc(m) for m in layers]
    sls = [SlicedGemv(m, rows_per_wave=a.rpw, exact=a.exact) for m in layers]
    x = torch.randn(1, 1, I, device=dev).to(dt)
    y = torch.empty(1, 1, O, device=dev, dtype=dt)
    layers[0].enable_sliced_layout(False)   # (the module's own one-token route would be the sliced one)
    ref = layers[0](x)
    got = sls[0](x)
    err = ((got.float() - ref.float()).abs().max() / ref.float().abs().max()).item()

    def run_default():
        for d, _ in descs:
            B.check(lib.vptq_quant_gemv(d, x.data_ptr(), y.data_ptr(), 1, B.GEMV_EXACT if a.exact else 0, None, 0, torch.cuda.current_stream().cuda_stream), "gemv")

    def run_sliced():
        for s in sls:
            s(x, y)
    us_d = time_graph(run_default, 10) / R
    us_s = time_graph(run_sliced, 10) / R
    idx_bytes = layers[0].indices.numel() * 4
    r = dict(I=I, O=O, v=a.v, k=a.k, kr=a.kr, arithmetic="reference roundings" if a.exact else "folded", dtype="bf16" if a.bf16 else "f16", default_us=us_d, default_kernel=lib.vptq_quant_gemv_kernel_name(descs[0][0], 1, 0).decode(),
             sliced_us=us_s, speedup=us_d / us_s, rel_diff=err, packed_index_MiB=idx_bytes / 2**20,
             layout_MiB=sls[0].extra_bytes / 2**20, rows_per_wave=sls[0].layout[0].rows_per_wave, slices=sls[0].slices,
             sliced_GBps_of_packed_bytes=idx_bytes / us_s / 1e3, sliced_GBps_of_layout_bytes=sls[0].extra_bytes / us_s / 1e3)
    print(json.dumps(r), flush=True); res.append(r)
    del layers, descs, sls
if a.out:
    json.dump(res, open(a.out, "w"), indent=1)
