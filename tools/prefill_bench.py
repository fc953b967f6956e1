#!/usr/bin/env python3
"""BASELINE config #4: many tokens (prefill: batch 4 x seq 2048 = 8192) through one VQuantLinear,
two routes, fp16 and bf16:
  * fused   : vptq_quant_gemm - dequantised tile -> LDS -> 32x32x16 MFMA (gemm_fused.hip)
  * dense   : HIP dequant to a dense W + hipBLASLt GEMM (torch F.linear) - the reference's structure
    python tools/prefill_bench.py --tokens 64,128,256,512,1024,2048,8192 --dtypes f16,bf16"""
import argparse, json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
from vptq_amd import _backend as B, ops  # noqa
from _gpu_util import module_desc  # noqa


def t_us(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def mk(I, O, dev, dt):
    import vptq_amd
    g = torch.Generator(device=dev).manual_seed(1)
    m = vptq_amd.VQuantLinear(I, O, vector_lens=[-1, 8], num_centroids=[-1, 256], num_res_centroids=[-1, 256],
                              group_num=1, group_size=I, outlier_size=0, indices_as_float=False, enable_norm=True,
                              enable_perm=False, is_indice_packed=True, bias=False, dtype=dt, device=dev,
                              enable_proxy_error=False)
    m.indices.data = torch.randint(-2**31, 2**31 - 1, m.indices.shape, generator=g, device=dev, dtype=torch.int64).to(torch.int32)
    m.centroids.weight.data = (torch.randn(m.centroids.weight.shape, generator=g, device=dev) * 0.02).to(dt)
    m.res_centroids.weight.data = (torch.randn(m.res_centroids.weight.shape, generator=g, device=dev) * 0.005).to(dt)
    m.weight_scale.data = (1 + 0.1 * torch.randn(I, generator=g, device=dev)).to(dt)
    m.weight_bias.data = (0.01 * torch.randn(I, generator=g, device=dev)).to(dt)
    return m.eval()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tokens", default="8192")
    ap.add_argument("--shapes", default="4096,4096;8192,8192;4096,14336")
    ap.add_argument("--dtypes", default="f16,bf16")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    res = []
    for dtn in a.dtypes.split(","):
        dt = torch.float16 if dtn == "f16" else torch.bfloat16
        for (I, O) in [tuple(int(v) for v in p.split(",")) for p in a.shapes.split(";")]:
            m = mk(I, O, dev, dt)
            desc, keep = module_desc(m)
            for T in [int(t) for t in a.tokens.split(",")]:
                x = torch.randn(1, T, I, device=dev, dtype=dt)
                dq = t_us(lambda: m.dequant())
                W = m.dequant()
                mm = t_us(lambda: torch.nn.functional.linear(x, W))
                dense = t_us(lambda: torch.nn.functional.linear(x, m.dequant()))
                fused = t_us(lambda: ops.quant_gemm_fused(x, desc, O))
                y_f, y_d = ops.quant_gemm_fused(x, desc, O), torch.nn.functional.linear(x, W)
                err = ((y_f.float() - y_d.float()).abs().max() / y_d.float().abs().max()).item()
                flops = 2.0 * T * I * O
                r = dict(dtype=dtn, I=I, O=O, tokens=T, fused_us=fused, fused_TFLOPs=flops / fused / 1e6,
                         fused_frac_of_2500TF=flops / fused / 1e6 / 2500, dequant_us=dq, gemm_us=mm,
                         dense_us=dense, dense_TFLOPs=flops / dense / 1e6, fused_vs_dense=dense / fused,
                         rel_diff=err)
                res.append(r)
                print(json.dumps(r), flush=True)
            del m, desc, keep
    if a.out:
        json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
