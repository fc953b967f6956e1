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
    ws = torch.zeros(256 * 16 * 16, dtype=torch.int64, device=dev)
    for rep in range(3):
        ws.zero_()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        B.check(B.lib().vptq_quant_gemv_chain(descs, n, xp, yp, 1, 0, ws.data_ptr(), ws.numel() * 8,
                                              B.current_stream_ptr(dev)), "chain")
        e1.record()
        torch.cuda.synchronize()
    w = ws.view(256, 16, 16).cpu().double()
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
    print(f"  steps per wave {steps:.1f}; slowest wave total {w[..., 5].max().item():.0f}, fastest {w[..., 5].min().item():.0f}")


if __name__ == "__main__":
    main()
