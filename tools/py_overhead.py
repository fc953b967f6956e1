import sys, time, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import vptq_amd
from shape_bench import mk
dev = torch.device("cuda", 0); g = torch.Generator(device=dev).manual_seed(0)
m = mk(4096, 4096, dev, g)
lin = torch.nn.Linear(4096, 4096, bias=False, device=dev, dtype=torch.float16)
x = torch.randn(1, 1, 4096, device=dev, dtype=torch.float16)
with torch.no_grad():
    for f, name in ((m, "VQuantLinear.forward"), (lin, "nn.Linear.forward")):
        for _ in range(200): f(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5000): f(x)
        t1 = time.perf_counter()      # enqueue time only (the queue never fills in 5000 x ~5 us)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"{name}: {(t1 - t0) / 5000 * 1e6:.1f} us per call to enqueue, {(t2 - t0) / 5000 * 1e6:.1f} us per call incl. drain")
import cProfile, pstats
with torch.no_grad():
    pr = cProfile.Profile(); pr.enable()
    for _ in range(2000): m(x)
    pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)

# sibling group (q / k / v through one grouped launch): Python cost of the three forward calls
class Blk(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.q_proj, self.k_proj, self.v_proj = mk(4096, 4096, dev, g), mk(4096, 1024, dev, g), mk(4096, 1024, dev, g)
blk = Blk()
assert vptq_amd.layers.link_siblings(blk) == 1
with torch.no_grad():
    def qkv():
        blk.q_proj(x); blk.k_proj(x); blk.v_proj(x)
    for _ in range(200): qkv()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3000): qkv()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"q + k + v through a sibling group: {(t1 - t0) / 3000 * 1e6:.1f} us per triple to enqueue")
    pr = cProfile.Profile(); pr.enable()
    for _ in range(1000): qkv()
    pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(10)
