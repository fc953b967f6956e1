#!/usr/bin/env python3
"""Phase timeline of gemv_k256m_kernel (GPU box only).

Needs the trace build of the library (every wave stamps the 100 MHz wall clock at its phase
boundaries into the buffer passed as the read-ahead range):

    make -C vptq_amd/csrc trace          # -> tools/_build/libvptq_hip_trace.so
    python tools/trace_k256m.py --hidden 8192 [--fast]

Prints, for one launch in the middle of a ring of cold layers, when (us after the first wave
of the launch started) the waves passed each boundary: min / median / max over all waves.
"""
import argparse
import ctypes as C
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from vptq_amd import _backend as B  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, "tests"))
import microbench as mb  # noqa: E402  (make_layers)
from _gpu_util import module_desc  # noqa: E402

PHASES = ["start", "first loads issued", "LDS image built", "accumulate done", "wave reduced", "after barrier",
          "activations arrived", "at the prologue barrier"]
ORDER = [0, 1, 6, 7, 2, 3, 4, 5]   # print in program order


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hidden", type=int, default=8192)
    ap.add_argument("--ring", type=int, default=0)
    ap.add_argument("--exact", action="store_true", help="VPTQ_GEMV_EXACT instead of the default arithmetic")
    ap.add_argument("--fast", action="store_true", help="accepted, no effect: the default")
    ap.add_argument("--hot", action="store_true", help="same layer every launch")
    ap.add_argument("--kernel", default="mfma", choices=["mfma", "valu"])
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    os.environ["VPTQ_K256_KERNEL"] = a.kernel
    os.environ["VPTQ_TUNING"] = "1"   # (tuning knobs are read only with it)
    dev = torch.device("cuda", 0)
    H = a.hidden
    lib = C.CDLL(os.environ.get("VPTQ_TRACE_LIB") or os.path.join(ROOT, "tools", "_build", "libvptq_hip_trace.so"))
    res, args = B.EXPORTS["vptq_quant_gemv"]
    lib.vptq_quant_gemv.restype, lib.vptq_quant_gemv.argtypes = res, args
    lib.vptq_last_error.restype = C.c_char_p
    R = a.ring or max(2, (512 << 20) // (H // 8 * H * 2))
    layers = mb.make_layers(H, R, dev)
    x = torch.randn(1, 1, H, device=dev, dtype=torch.float16)
    # workgroup geometry of the kernel traced: (rows per workgroup, waves per workgroup)
    rows, nw = (4, 16) if a.kernel == "mfma" else ((2 if H // 16 >= 512 else 1), 8)
    n_wg = (H // 8 + rows - 1) // rows
    bufs = [torch.zeros(n_wg * nw * 8, dtype=torch.int64, device=dev) for _ in range(R)]
    descs, keeps = [], []
    for i, m in enumerate(layers):
        d, k = module_desc(layers[0] if a.hot else m, prefetch=bufs[i])
        descs.append(d); keeps.append(k)
    y = torch.empty(1, 1, H, device=dev, dtype=torch.float16)
    st = torch.cuda.current_stream().cuda_stream
    flags = 4 if a.exact else 0
    for rep in range(3):
        for i in range(R):
            rc = lib.vptq_quant_gemv(descs[i], x.data_ptr(), y.data_ptr(), 1, flags, None, 0, st)
            assert rc == 0, lib.vptq_last_error()
    torch.cuda.synchronize()
    out = dict(hidden=H, ring=R, exact=a.exact, hot=a.hot, phases={})
    mid = R // 2
    t = bufs[mid].view(n_wg * nw, 8).cpu().double() * 0.01  # us
    prev_end = bufs[mid - 1].view(n_wg * nw, 8)[:, 5].cpu().double().max().item() * 0.01
    t0 = t[:, 0].min().item()
    print(f"{a.kernel} H={H} ring={R} exact={a.exact} hot={a.hot}: previous launch's last wave ended "
          f"{t0 - prev_end:+.2f} us before this launch's first wave started")
    for k in ORDER:
        name = PHASES[k]
        if (t[:, k] == 0).all() or (k == 4 and a.kernel == "mfma"):
            continue  # boundary not stamped by this kernel
        v = (t[:, k] - t0)
        q = [v.min().item(), v.median().item(), v.max().item()]
        # the slowest wave of each workgroup (what a barrier or the end of the kernel waits for)
        w = v.view(n_wg, nw).max(dim=1).values
        out["phases"][name] = q + [w.min().item(), w.median().item(), w.max().item()]
        print(f"  {name:24s} min {q[0]:6.2f}  median {q[1]:6.2f}  max {q[2]:6.2f} us   slowest wave per "
              f"workgroup: {w.min().item():5.2f} / {w.median().item():5.2f} / {w.max().item():5.2f}")
    if a.kernel == "mfma":
        # slot 4 of the MFMA kernel: shader-clock (s_memtime) cycles between "LDS image built" and "accumulate done"
        cyc = t[:, 4] / 0.01
        dt = t[:, 3] - t[:, 2]
        ok = dt > 0
        mhz = (cyc[ok] / dt[ok])
        out["accumulate_clock_MHz"] = [mhz.min().item(), mhz.median().item(), mhz.max().item()]
        print(f"  s_memtime rate during the accumulate phase: min {mhz.min().item():.0f}  median {mhz.median().item():.0f}  "
              f"max {mhz.max().item():.0f} counts per us; accumulate phase median {cyc[ok].median().item():.0f} counts")
    nxt = bufs[mid + 1].view(n_wg * nw, 8)[:, 0].cpu().double().min().item() * 0.01
    out["launch_period_us"] = nxt - t0
    print(f"  next launch's first wave started {nxt - t0:.2f} us after this one's")
    if a.out:
        json.dump(out, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
