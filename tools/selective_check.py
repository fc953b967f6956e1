# quick GPU check of VPTQ_GEMV_SELECTIVE in the chain launch: parity against dequant + float64, per activation kind
import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
import bench
from vptq_amd import _backend as B
from vptq_amd.ops.chain import GemvChain
import gpu_gate_count as gc
dev = torch.device("cuda", 0)
B.set_arithmetic("folded")
g = torch.Generator(device=dev).manual_seed(5)
H = 8192
for fam in ('ckpt', 'llm-r4', 'outlier-cols', 'big-bias'):
    ring = [gc.make(H, H, fam, torch.float16, dev, g) for _ in range(12)]
    ch = GemvChain(ring)
    for xk in gc.XKINDS:
        xs = [gc.make_x(m, xk, torch.float16, dev, g) for m in ring]
        res = {}
        for name, fl in (('exact', B.GEMV_EXACT), ('sel', B.GEMV_SELECTIVE), ('folded', 0)):
            ys = ch(xs, flags=fl | B.GEMV_FORCE_MFMA)
            torch.cuda.synchronize()
            worst = 0.0
            for m, x, y in zip(ring, xs, ys):
                W = m.dequant()
                r16 = (W.double() @ x.reshape(-1).double()).half().double()
                worst = max(worst, float((y.reshape(-1).double() - r16).abs().max() / r16.abs().max()))
            res[name] = worst
        print(fam, xk, {k: f'{v:.2e}' for k, v in res.items()}, ch.kernel_name(flags=B.GEMV_SELECTIVE | B.GEMV_FORCE_MFMA), flush=True)
    del ring, ch
    torch.cuda.empty_cache()
