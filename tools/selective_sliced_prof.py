#!/usr/bin/env python3
"""One two-table large-codebook layer, one token, in the three arithmetics over the sliced layouts (rocprofv3 --kernel-trace --stats
-- python tools/selective_sliced_prof.py gives the kernels' own durations: gemv_hot_kernel, gemv_sliced_kernel<...>)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vptq_amd.utils.sliced import SlicedGemv
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
H = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
v = int(sys.argv[2]) if len(sys.argv) > 2 else 8
layers = [bench.make_layer(H, H, dev, g, 65536, 65536, v=v) for _ in range(4)]
x = torch.randn(1, 1, H, device=dev, generator=g).half()
objs = {"selective": [SlicedGemv(m, selective=True) for m in layers], "folded": [SlicedGemv(m) for m in layers],
        "reference": [SlicedGemv(m, exact=True) for m in layers]}
for name, ol in objs.items():
    for _ in range(3):
        for o in ol:
            o(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        for o in ol:
            o(x)
    e1.record()
    torch.cuda.synchronize()
    print(f"{name:10s} {e0.elapsed_time(e1) * 1e3 / 200:.2f} us per layer (eager launches, events)")
