#!/usr/bin/env python3
"""BASELINE config #4: prefill (batch 4 x seq 2048 = 8192 tokens) through one VQuantLinear:
HIP dequant-to-dense + hipBLASLt GEMM (what ops.quant_gemm does above 8 tokens)."""
import argparse, json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from microbench import make_layers  # noqa


def t_us(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tokens", type=int, default=8192)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    res = []
    for (I, O) in ((4096, 4096), (4096, 14336), (8192, 8192)):
        import vptq_amd
        m = make_layers(I, 1, dev)[0] if I == O else None
        if m is None:
            g = torch.Generator(device=dev).manual_seed(1)
            m = vptq_amd.VQuantLinear(I, O, vector_lens=[-1, 8], num_centroids=[-1, 256], num_res_centroids=[-1, 256],
                                      group_num=1, group_size=I, outlier_size=0, indices_as_float=False, enable_norm=True,
                                      enable_perm=False, is_indice_packed=True, bias=False, dtype=torch.float16, device=dev,
                                      enable_proxy_error=False)
            m.indices.data = torch.randint(-2**31, 2**31 - 1, m.indices.shape, generator=g, device=dev, dtype=torch.int64).to(torch.int32)
            m.centroids.weight.data = (torch.randn(m.centroids.weight.shape, generator=g, device=dev) * 0.02).half()
            m.res_centroids.weight.data = (torch.randn(m.res_centroids.weight.shape, generator=g, device=dev) * 0.005).half()
            m.weight_scale.data = (1 + 0.1 * torch.randn(I, generator=g, device=dev)).half()
            m.weight_bias.data = (0.01 * torch.randn(I, generator=g, device=dev)).half()
        x = torch.randn(4, a.tokens // 4, I, device=dev, dtype=torch.float16)
        W = m.dequant()
        dq = t_us(lambda: m.dequant())
        mm = t_us(lambda: torch.nn.functional.linear(x, W))
        full = t_us(lambda: m(x))
        flops = 2.0 * a.tokens * I * O
        res.append(dict(I=I, O=O, tokens=a.tokens, dequant_us=dq, dequant_write_TBps=2.0 * I * O / dq / 1e6,
                        gemm_us=mm, gemm_TFLOPs=flops / mm / 1e6, forward_us=full,
                        forward_TFLOPs=flops / full / 1e6, mfma_frac_of_2500TF=flops / full / 1e6 / 2500))
        print(json.dumps(res[-1]), flush=True)
    if a.out:
        json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
