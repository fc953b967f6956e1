#!/usr/bin/env python3
"""Does this torch / RCCL pair capture an all-reduce into a hipGraph?  (GPU box; world size 1 is enough to
go through RCCL's enqueue + capture path.)  bench.py --gpus N > 1 captures its token step - fused GEMVs and
RCCL all-reduces - into one graph; a refused capture cannot be recovered from in-process on this runtime."""
import os
import time

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
part = torch.randn(4, 8192, device=dev)
y = torch.empty(4, 8192, device=dev, dtype=torch.float16)
s = torch.cuda.Stream()


def one_pass():
    for _ in range(8):
        part.mul_(1.0001)
        dist.all_reduce(part)
        y.copy_(part)


with torch.cuda.stream(s):
    one_pass()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        one_pass()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        g.replay()
    torch.cuda.synchronize()
    tg = (time.perf_counter() - t0) / 50
    t0 = time.perf_counter()
    for _ in range(50):
        one_pass()
    torch.cuda.synchronize()
    te = (time.perf_counter() - t0) / 50
print(f"RCCL all-reduce captured and replayed: graph {tg * 1e6 / 8:.1f} us per (scale + all-reduce + copy), eager {te * 1e6 / 8:.1f} us; "
      f"torch {torch.__version__}, nccl {torch.cuda.nccl.version()}")
dist.destroy_process_group()
