#!/usr/bin/env python3
"""BASELINE config #3: Llama-3-8B-shaped decode loop on ONE MI355X with every decoder
nn.Linear replaced by a 2-bit VQuantLinear (random indices / centroids — there is no
network for checkpoints), through Hugging Face Transformers' own VPTQ integration:

    transformers.integrations.vptq.replace_with_vptq_linear  ->  `from vptq import VQuantLinear`

which resolves to this repository's class (the `vptq` alias package).  Reports TTFT
(prompt through dequant + hipBLASLt GEMM) and decode tokens/s, eager and with the decode step
captured in a hipGraph.

    python tools/llama_decode.py [--layers 32] [--prompt 128] [--new 256]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


SIZES = {"8b": dict(hidden_size=4096, intermediate_size=14336, num_attention_heads=32, layers=32),
         "70b": dict(hidden_size=8192, intermediate_size=28672, num_attention_heads=64, layers=80)}


def build_model(layers, dev, k=256, kr=256, perm=False, size="8b", dt=torch.float16):
    from transformers import LlamaConfig, LlamaForCausalLM
    from transformers.integrations.vptq import replace_with_vptq_linear
    from transformers.models.llama.modeling_llama import LlamaRotaryEmbedding
    from transformers.utils.quantization_config import VptqConfig
    import vptq  # the alias package of this repository
    assert vptq.__file__.startswith(ROOT), vptq.__file__

    sz = SIZES[size]
    cfg = LlamaConfig(hidden_size=sz["hidden_size"], intermediate_size=sz["intermediate_size"],
                      num_hidden_layers=layers or sz["layers"],
                      num_attention_heads=sz["num_attention_heads"], num_key_value_heads=8, vocab_size=128256,
                      max_position_embeddings=8192, rope_theta=500000.0, rms_norm_eps=1e-5,
                      tie_word_embeddings=False)
    cfg._attn_implementation = "sdpa"
    with torch.device("meta"):
        model = LlamaForCausalLM(cfg)
    per_layer = {}
    for name, mod in model.named_modules():
        if isinstance(mod, torch.nn.Linear) and name != "lm_head":
            per_layer[name] = dict(vector_lens=[-1, 8], num_centroids=[-1, k],
                                   num_res_centroids=[-1, kr if kr > 0 else -1], group_num=1,
                                   group_size=mod.in_features, outlier_size=0,
                                   indices_as_float=False, enable_norm=True, enable_perm=perm,
                                   is_indice_packed=True)
    qc = VptqConfig(config_for_layers=per_layer, shared_layer_config={},
                    modules_to_not_convert=["lm_head"])
    try:
        replace_with_vptq_linear(model, modules_to_not_convert=["lm_head"], quantization_config=qc)
    except KeyError:
        # transformers 5.15.0 indexes `model._modules[<dotted name>]` (integrations/vptq.py:71)
        # and fails on any nested model; do exactly what it would have done: build
        # `vptq.VQuantLinear` on meta with the same keyword arguments and swap it in.
        for name, lp in per_layer.items():
            old = model.get_submodule(name)
            with torch.device("meta"):
                new = vptq.VQuantLinear(
                    old.in_features, old.out_features, vector_lens=lp["vector_lens"],
                    num_centroids=lp["num_centroids"], num_res_centroids=lp["num_res_centroids"],
                    group_num=lp["group_num"], group_size=lp["group_size"],
                    outlier_size=lp["outlier_size"], indices_as_float=lp["indices_as_float"],
                    enable_norm=lp["enable_norm"], enable_perm=lp["enable_perm"],
                    is_indice_packed=True, enable_proxy_error=False, bias=old.bias is not None)
            model.set_submodule(name, new)
    model = model.to_empty(device=dev)
    g = torch.Generator(device=dev).manual_seed(0)
    qlayers = []
    with torch.no_grad():
        for name, mod in model.named_modules():
            if isinstance(mod, vptq.VQuantLinear):
                mod.indices.data = torch.randint(-2**31, 2**31 - 1, mod.indices.shape, generator=g,
                                                 device=dev, dtype=torch.int64).to(torch.int32)
                mod.centroids.weight.data = (torch.randn(mod.centroids.weight.shape, generator=g, device=dev) * 0.02).to(dt)
                if mod.enable_residual:
                    mod.res_centroids.weight.data = (torch.randn(mod.res_centroids.weight.shape, generator=g, device=dev) * 0.005).to(dt)
                I = mod.in_features
                mod.weight_scale.data = (1 + 0.1 * torch.randn(I, generator=g, device=dev)).to(dt)
                mod.weight_bias.data = (0.002 * torch.randn(I, generator=g, device=dev)).to(dt)
                if perm:
                    mod.perm.data = torch.randperm(I, generator=g, device=dev).to(torch.int32).to(torch.int16)
                qlayers.append(mod)
        for name, p in model.named_parameters():
            if p.dtype == torch.float32:          # embeddings, norms, lm_head
                if "norm" in name:
                    p.data = torch.ones(p.shape, device=dev, dtype=dt)
                else:
                    p.data = (torch.randn(p.shape, generator=g, device=dev) * 0.02).to(dt)
    model.model.rotary_emb = LlamaRotaryEmbedding(config=cfg, device=dev)
    vptq.layers.chain_prefetch(qlayers, circular=True)
    return model.eval(), cfg, qlayers


@torch.no_grad()
def run(args):
    from transformers import StaticCache
    dev = torch.device("cuda", 0)
    def stage(msg):
        print(f"[stage] {msg}", file=sys.stderr, flush=True)
    stage("build")
    model, cfg, qlayers = build_model(args.layers, dev, k=args.k, kr=args.kr, perm=args.perm, size=args.model, dt=torch.bfloat16 if args.dtype == "bf16" else torch.float16)
    fused = 0
    if args.fuse:
        import vptq
        fused = vptq.layers.link_siblings(model)
        stage(f"{fused} sibling groups linked")
    torch.cuda.synchronize()
    stage("built")
    qbytes = sum(m.indices.numel() * 4 for m in qlayers)
    maxlen = args.prompt + 2 * args.new + 80   # eager steps + captured replays share one cache
    prompt = torch.randint(0, cfg.vocab_size, (args.batch, args.prompt), device=dev)

    def fresh_cache():
        return StaticCache(config=cfg, max_cache_len=maxlen)

    # ---- TTFT: prompt (tokens > 8 -> dequant + F.linear per layer) ----
    cache = fresh_cache()
    pos = torch.arange(args.prompt, device=dev)
    stage("prompt forward")
    out = model(input_ids=prompt, past_key_values=cache, cache_position=pos, use_cache=True)
    torch.cuda.synchronize()
    stage("prompt done")
    cache = fresh_cache()
    t0 = time.perf_counter()
    out = model(input_ids=prompt, past_key_values=cache, cache_position=pos, use_cache=True)
    tok = out.logits[:, -1].argmax(-1, keepdim=True)
    torch.cuda.synchronize()
    ttft = time.perf_counter() - t0

    # ---- eager decode ----
    def step(tok, p):
        o = model(input_ids=tok, past_key_values=cache, cache_position=p, use_cache=True)
        return o.logits[:, -1].argmax(-1, keepdim=True)

    p = torch.tensor([args.prompt], device=dev)
    stage("eager decode")
    for _ in range(3):
        tok = step(tok, p); p += 1
    torch.cuda.synchronize()
    n_eager = min(args.new, 64)
    t0 = time.perf_counter()
    for _ in range(n_eager):
        tok = step(tok, p); p += 1
    torch.cuda.synchronize()
    eager_tps = n_eager * args.batch / (time.perf_counter() - t0)

    # ---- hipGraph decode: one captured step, replayed ----
    graph_tps = None
    stage("graph capture")
    try:
        s_tok = tok.clone()
        s_pos = p.clone()
        stream = torch.cuda.Stream()
        with torch.cuda.stream(stream):
            for _ in range(2):
                s_out = step(s_tok, s_pos)
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=stream):
                s_out = step(s_tok, s_pos)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.new):
                gr.replay()
                s_tok.copy_(s_out)
                s_pos += 1
            torch.cuda.synchronize()
            graph_tps = args.new * args.batch / (time.perf_counter() - t0)
    except Exception as e:  # report, do not hide
        graph_tps = f"capture failed: {type(e).__name__}: {e}"

    # ---- the VQuantLinear share of a decode step: the step's own VQuantLinear calls (same modules, same order, same
    # activation objects - so sibling groups launch together as in the model), alone in a hipGraph of their own
    vq_us = None
    stage("VQuantLinear share")
    try:
        import vptq
        calls, hooks = [], []
        for m in qlayers:
            hooks.append(m.register_forward_pre_hook(lambda mod, inp: calls.append((mod, inp[0]))))
        step(tok, p)
        for h in hooks:
            h.remove()
        torch.cuda.synchronize()
        stream2 = torch.cuda.Stream()
        with torch.cuda.stream(stream2):
            def only_vq():
                for mod, xin in calls:
                    mod(xin)
            for _ in range(2):
                only_vq()
            torch.cuda.synchronize()
            g2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g2, stream=stream2):
                only_vq()
            for _ in range(5):
                g2.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(50):
                g2.replay()
            torch.cuda.synchronize()
            vq_us = (time.perf_counter() - t0) * 1e6 / 50
    except Exception as e:  # report, do not hide
        vq_us = f"failed: {type(e).__name__}: {e}"

    lm_head_bytes = cfg.vocab_size * cfg.hidden_size * 2
    res = dict(model=f"Llama-3-{args.model.upper()} shapes, {cfg.num_hidden_layers} layers, VQuantLinear v8-k{args.k}-{args.kr}" + (" (2-bit)" if (args.k, args.kr) == (256, 256) else "")
                     + (" +perm" if args.perm else ""),
               quantized_linears=len(qlayers), packed_index_GB=qbytes / 1e9,
               lm_head_GB=lm_head_bytes / 1e9, prompt=args.prompt, new_tokens=args.new,
               ttft_ms=ttft * 1e3, decode_tok_s_eager=eager_tps, decode_tok_s_hipgraph=graph_tps,
               sibling_groups=fused, batch=args.batch)
    if isinstance(graph_tps, float):
        res["hipgraph_weight_GBps"] = (qbytes + lm_head_bytes) * graph_tps / args.batch / 1e9   # (a step reads the weights once for the whole batch)
    res["vqlinear_us_per_token"] = vq_us   # the step's VQuantLinear launches back to back in a graph of their own (batch > 1: per STEP)
    if isinstance(vq_us, float):
        res["vqlinear_calls_per_token"] = len(calls)
        res["vqlinear_GBps"] = qbytes / vq_us / 1e3
        if isinstance(graph_tps, float):
            res["vqlinear_share_of_step"] = vq_us * 1e-6 * graph_tps / args.batch
    print(json.dumps(res))
    if args.out:
        os.makedirs(os.path.dirname(args.out), exist_ok=True)
        json.dump(res, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="8b", choices=sorted(SIZES))
    ap.add_argument("--layers", type=int, default=0, help="0 = the model's own depth")
    ap.add_argument("--prompt", type=int, default=128)
    ap.add_argument("--new", type=int, default=256)
    ap.add_argument("--perm", action="store_true")
    ap.add_argument("--k", type=int, default=256, help="main codebook entries (v = 8): 256 = the 2-bit format, 65536 = the published 3-bit ones")
    ap.add_argument("--kr", type=int, default=256, help="residual codebook entries")
    ap.add_argument("--fuse", action="store_true", help="link_siblings: q/k/v and gate/up share one grouped launch")
    ap.add_argument("--batch", type=int, default=1, help="sequences decoded together: every VQuantLinear call of a step sees that many tokens")
    ap.add_argument("--dtype", default="f16", choices=["f16", "bf16"], help="dtype of the model and its VQuantLinear tensors")
    ap.add_argument("--out", default="")
    run(ap.parse_args())
