#!/usr/bin/env python3
"""Per-wave cycle accounting of the chain kernel (library built with -DVPTQ_K256C_PROF=1, run with
VPTQ_K256C_PROF=1): one launch of `--layers` layers, averages over all waves.
  VPTQ_HIP_LIB=tools/_build/libvptq_hip_prof.so VPTQ_K256C_PROF=1 python tools/chain_prof.py --hidden 8192"""
import argparse, ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hidden", type=int, default=8192)
    ap.add_argument("--rows", type=int, default=0)
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--waves", type=int, default=16, help="waves per workgroup of the build (VPTQ_K256C_WAVES)")
    ap.add_argument("--timeline", action="store_true", help="library built with -DVPTQ_K256C_PROF=2: who computes when")
    a = ap.parse_args()
    import bench
    from vptq_amd import _backend as B
    dev = torch.device("cuda", 0)
    I, O, n = a.hidden, a.rows or a.hidden, a.layers
    g = torch.Generator(device=dev).manual_seed(1)
    ring = [bench.make_layer(I, O, dev, g) for _ in range(n)]
    x = torch.randn(1, 1, I, device=dev, generator=g).half()
    ys = [torch.empty(1, 1, O, dtype=torch.float16, device=dev) for _ in range(n)]
    descs = (B.LayerDesc * n)(*[m._descriptor()[1] for m in ring])
    xp = (C.c_void_p * n)(*[x.data_ptr()] * n)
    yp = (C.c_void_p * n)(*[y.data_ptr() for y in ys])
    ws = torch.zeros(256 * 16 * 64, dtype=torch.int64, device=dev)
    for rep in range(3):
        ws.zero_()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        B.check(B.lib().vptq_quant_gemv_chain(descs, n, xp, yp, 1, 0, ws.data_ptr(), ws.numel() * 8,
                                              B.current_stream_ptr(dev)), "chain")
        e1.record()
        torch.cuda.synchronize()
    raw = ws.view(256 * 16 * 64)[:256 * a.waves * 64].view(256, a.waves, 64).cpu()
    w = raw.double()
    us = e0.elapsed_time(e1) * 1e3
    tot = w[..., 5].mean().item()
    print(f"launch {us:.1f} us = {us / n:.2f} us per layer; wave total {tot:.0f} clocks of s_memtime "
          f"({tot / us:.1f} per us)")
    names = ["wait for index words", "consume", "request next (+ fill request, issue-side row group / layer switch)",
             "rare paths (sums, consume-side row group / layer switch)"]
    steps = w[..., 4].mean().item()
    for i, nm in enumerate(names):
        v = w[..., i].mean().item()
        print(f"  {nm:72s} {v:12.0f}  {100 * v / tot:5.1f} %   per step {v / steps:8.1f}   per layer {v / n:9.1f}")
    sub = ["sums: deposit", "sums: store the previous row group (wave q mod 16)", "sums: store the layer's last row group",
           "leave layer: free + pending fill", "enter layer: arguments", "enter layer: wait for the image",
           "enter layer: find the next layer", "issue side: find + load the next layer"]
    for i, nm in enumerate(sub):
        v = w[..., 8 + i].mean().item()
        print(f"    {nm:70s} {v:12.0f}  {100 * v / tot:5.1f} %   per layer {v / n:9.1f}   worst wave {w[..., 8 + i].max().item():9.0f}")
    pro = w[..., 6]
    ent = w[..., 7]
    live = w[..., 4] > 0
    if live.any():
        e0 = ent[live].min().item()
        print(f"  prologue (kernel entry -> first step) per wave: mean {pro[live].mean().item():.0f}, max {pro[live].max().item():.0f} clocks; "
              f"entry stamps spread over {ent[live].max().item() - e0:.0f} clocks; last wave ends {(ent[live] + pro[live] + w[..., 5][live]).max().item() - e0:.0f} clocks after the first entry")
    if live.any():
        names2 = ["layer search (kernel-argument loads)", "layer record loaded", "row group set up + counters zeroed + barrier",
                  "image DMA + first sweeps requested", "image + first sweeps' older loads landed", "next layer planned (2 more argument loads)",
                  "every wave's image part landed"]
        for i, nm in enumerate(names2):
            v = w[..., 16 + i][live]
            print(f"    prologue mark {i}: {nm:58s} mean {v.mean().item():8.0f}  max {v.max().item():8.0f} clocks since entry")
    if a.timeline:
        timeline(raw, a.waves)
    print(f"  steps per wave {steps:.1f}; slowest wave total {w[..., 5].max().item():.0f}, fastest {w[..., 5].min().item():.0f}")


def timeline(raw, waves):
    """raw[wg, wave, 24 + i] = consume start | end << 32 (low 32 bits of s_memtime) of steps 16 .. 47; [.., 23] = HW_ID.
    For every SIMD of every workgroup: how many of its waves are inside a consume phase, sampled over the window all of
    them cover; and the same per CU (LDS)."""
    import numpy as np
    r = raw.numpy()
    hw = r[..., 23]
    simd = (hw >> 4) & 3
    st = (r[..., 24:56] & 0xffffffff).astype(np.int64)
    en = ((r[..., 24:56] >> 32) & 0xffffffff).astype(np.int64)
    hist_simd = np.zeros(waves + 1)
    hist_cu = np.zeros(waves + 1)
    shown = 0
    for wg in range(r.shape[0]):
        if (st[wg] == 0).any():
            continue
        base = st[wg].min()
        s_, e_ = st[wg] - base, en[wg] - base
        if (e_ < s_).any() or e_.max() > 1 << 30:
            continue   # (the 32-bit stamps wrapped inside this window)
        lo, hi = s_[:, 0].max(), e_[:, -1].min()
        if hi <= lo:
            continue
        ts = np.arange(lo, hi, 16)
        inside = ((ts[None, None, :] >= s_[:, :, None]) & (ts[None, None, :] < e_[:, :, None])).any(axis=1)   # [wave, t]
        cnt = inside.sum(axis=0)
        hist_cu += np.bincount(cnt, minlength=waves + 1)[:waves + 1]
        for sd in range(4):
            m = simd[wg] == sd
            if m.any():
                c = inside[m].sum(axis=0)
                hist_simd += np.bincount(c, minlength=waves + 1)[:waves + 1]
        if shown < 2:
            shown += 1
            print(f"  timeline of workgroup {wg} (one character = 64 clocks; digit = SIMD of the wave while it consumes, '.' otherwise):")
            t2 = np.arange(lo, min(hi, lo + 64 * 150), 64)
            for wv in np.argsort(simd[wg], kind="stable"):
                ins = ((t2[None, :] >= s_[wv][:, None]) & (t2[None, :] < e_[wv][:, None])).any(axis=0)
                print("    w%02d simd %d  " % (wv, simd[wg][wv]) + "".join(str(simd[wg][wv]) if v else "." for v in ins))
    if hist_simd.sum() > 0:
        hs, hc = hist_simd / hist_simd.sum(), hist_cu / hist_cu.sum()
        print("  waves of a SIMD inside a consume phase at the same time: " +
              "  ".join(f"{i}: {100 * v:.1f} %" for i, v in enumerate(hs) if v > 0.0005) +
              f"   (mean {sum(i * v for i, v in enumerate(hs)):.2f})")
        print("  waves of a CU inside a consume phase at the same time:   " +
              "  ".join(f"{i}: {100 * v:.1f} %" for i, v in enumerate(hc) if v > 0.0005) +
              f"   (mean {sum(i * v for i, v in enumerate(hc)):.2f})")


if __name__ == "__main__":
    main()
