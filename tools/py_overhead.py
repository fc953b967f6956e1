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
