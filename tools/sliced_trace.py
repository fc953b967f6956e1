#!/usr/bin/env python3
"""Phase timeline of ONE launch of gemv_sliced_kernel from a -DVPTQ_SLICED_TRACE=1 build (every wave stamps s_memrealtime,
100 MHz, at entry / behind the prologue barrier / at the end of its stream / at its exit behind the accumulator words of
the workspace).  8192^2-sized layers (the stamps sit where the round-4 partial sums were).
    VPTQ_HIP_LIB=tools/_build/libvptq_hip_tr1.so python tools/sliced_trace.py --kr 0"""
import argparse, json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from vptq_amd.utils.sliced import SlicedGemv  # noqa
from shape_bench import mk  # noqa

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="8192,8192")
ap.add_argument("--kr", type=int, default=0)
ap.add_argument("--k", type=int, default=65536)
ap.add_argument("--v", type=int, default=8)
ap.add_argument("--ring", type=int, default=6)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--raw", default="", help="save the last run's stamps [workgroups, 16 waves, 4] (us from the first entry) as .npy")
a = ap.parse_args()
dev = torch.device("cuda", 0); g = torch.Generator(device=dev).manual_seed(0)
I, O = (int(v) for v in a.shape.split(","))
layers = [mk(I, O, dev, g, k=a.k, kr=a.kr, v=a.v) for _ in range(a.ring)]
sls = [SlicedGemv(m) for m in layers]
x = torch.randn(1, 1, I, device=dev).half()
y = torch.empty(1, 1, O, device=dev, dtype=torch.float16)
N = O // a.v
res = []
for rep in range(a.reps):
    for s in sls:            # HBM-cold ring: the traced launch is the LAST layer's, behind the others
        s(x, y)
    torch.cuda.synchronize()
    ws = next(iter(sls[-1]._ws.values()))
    words = ws.view(torch.int64)
    n_wg = None
    st = words[N * a.v:].cpu()
    # workgroups that ran have a non-zero entry stamp
    st = st[: (st.numel() // 64) * 64].reshape(-1, 16, 4)
    live = st[:, 0, 0] != 0
    st = st[live].double()
    t0 = st[:, :, 0].min()
    us = (st - t0) / 100.0   # 100 MHz -> us
    q = lambda t: [round(float(v), 2) for v in (t.min(), t.median(), t.max())]
    res.append(dict(workgroups=int(live.sum()), entry=q(us[:, :, 0]), prologue_done=q(us[:, :, 1]), stream_done=q(us[:, :, 2]),
                    exit=q(us[:, :, 3]), stream_len=q(us[:, :, 2] - us[:, :, 1]), tail=q(us[:, :, 3] - us[:, :, 2])))
    if a.raw and rep == a.reps - 1:
        import numpy as np
        np.save(a.raw, us.numpy())
    words[N * a.v:].zero_()
print(json.dumps(dict(shape=[I, O], v=a.v, k=a.k, kr=a.kr, lib=os.environ.get("VPTQ_HIP_LIB", "default"),
                      note="min / median / max over waves, us from the first wave's entry", runs=res), indent=1))
