#!/usr/bin/env python3
"""Diagnostic: every Llama-3-8B linear shape through the fused GEMV vs dequant + fp32 matmul."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vptq_amd
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
def mk(I, O, perm=False):
    m = vptq_amd.VQuantLinear(I, O, vector_lens=[-1, 8], num_centroids=[-1, 256], num_res_centroids=[-1, 256],
                              group_num=1, group_size=I, outlier_size=0, indices_as_float=False, enable_norm=True,
                              enable_perm=perm, is_indice_packed=True, bias=False, dtype=torch.float16, device=dev,
                              enable_proxy_error=False)
    m.indices.data = torch.randint(-2**31, 2**31 - 1, m.indices.shape, generator=g, device=dev, dtype=torch.int64).to(torch.int32)
    m.centroids.weight.data = (torch.randn(m.centroids.weight.shape, generator=g, device=dev) * 0.02).half()
    m.res_centroids.weight.data = (torch.randn(m.res_centroids.weight.shape, generator=g, device=dev) * 0.005).half()
    m.weight_scale.data = (1 + 0.1 * torch.randn(I, generator=g, device=dev)).half()
    m.weight_bias.data = (0.002 * torch.randn(I, generator=g, device=dev)).half()
    if perm:
        m.perm.data = torch.randperm(I, generator=g, device=dev).to(torch.int32).to(torch.int16)
    return m
shapes = [(4096, 4096), (4096, 1024), (4096, 14336), (14336, 4096)]
mods = [mk(I, O) for I, O in shapes]
for pf in (False, True):
    if pf:
        vptq_amd.layers.chain_prefetch(mods, circular=True)
    for (I, O), m in zip(shapes, mods):
        for tokens in (1, 2, 5, 128):
            x = torch.randn(1, tokens, I, device=dev, dtype=torch.float16, generator=g)
            print(f"prefetch={pf} I={I} O={O} tokens={tokens} ...", end="", flush=True)
            y = m(x)
            torch.cuda.synchronize()
            W = m.dequant()
            ref = x.float().reshape(tokens, I) @ W.float().t()
            err = ((y.float().reshape(tokens, O) - ref).abs().max() / ref.abs().max()).item()
            print(f" ok rel_err={err:.2e}", flush=True)
            assert err < 2e-3
print("ALL SHAPES OK", flush=True)
