"""Write a tiny synthetic VPTQ checkpoint (config.json + model.safetensors) to a directory."""
import json
import os

import torch


def write_tiny_checkpoint(path, hidden=256, inter=512, layers=2, heads=4, kv_heads=2, vocab=320,
                          seed=0, perm=True):
    import transformers
    from safetensors.torch import save_file
    import vptq_amd
    cfg = transformers.LlamaConfig(hidden_size=hidden, intermediate_size=inter,
                                   num_hidden_layers=layers, num_attention_heads=heads,
                                   num_key_value_heads=kv_heads, vocab_size=vocab,
                                   max_position_embeddings=256, tie_word_embeddings=False)
    with torch.device("meta"):
        model = transformers.LlamaForCausalLM(cfg)
    per_layer = {}
    for name, mod in model.named_modules():
        if isinstance(mod, torch.nn.Linear) and name != "lm_head":
            per_layer[name] = dict(in_features=mod.in_features, out_features=mod.out_features,
                                   vector_lens=[-1, 8], num_centroids=[-1, 256],
                                   num_res_centroids=[-1, 256], group_num=1,
                                   group_size=mod.in_features, outlier_size=0,
                                   indices_as_float=False, enable_norm=True, enable_perm=perm,
                                   is_indice_packed=True, bias=False)
    g = torch.Generator().manual_seed(seed)
    state = {}
    for name, p in model.state_dict().items():
        mod_name = name.rsplit(".", 1)[0]
        if mod_name in per_layer:
            continue
        if "norm" in name:
            state[name] = torch.ones(p.shape, dtype=torch.float16)
        else:
            state[name] = (torch.randn(p.shape, generator=g) * 0.05).half()
    for name, kw in per_layer.items():
        m = vptq_amd.VQuantLinear(**kw, dtype=torch.float16, enable_proxy_error=False)
        I = kw["in_features"]
        sd = {
            "indices": torch.randint(-2**31, 2**31 - 1, m.indices.shape, generator=g, dtype=torch.int64).to(torch.int32),
            "centroids.weight": (torch.randn(m.centroids.weight.shape, generator=g) * 0.05).half(),
            "res_centroids.weight": (torch.randn(m.res_centroids.weight.shape, generator=g) * 0.01).half(),
            "weight_scale": (1 + 0.1 * torch.randn(I, generator=g)).half(),
            "weight_bias": (0.01 * torch.randn(I, generator=g)).half(),
        }
        if perm:
            sd["perm"] = torch.randperm(I, generator=g).to(torch.int32).to(torch.int16)
        for k, v in sd.items():
            state[f"{name}.{k}"] = v.contiguous()
    os.makedirs(path, exist_ok=True)
    save_file(state, os.path.join(path, "model.safetensors"))
    cd = cfg.to_dict()
    cd["architectures"] = ["LlamaForCausalLM"]
    cd["dtype"] = "float16"
    cd["quantization_config"] = {"quant_method": "vptq", "config_for_layers": per_layer,
                                 "shared_layer_config": {}}
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(cd, f)
    return state, per_layer


def write_tiny_tokenizer(path, vocab=320, chat_template=True):
    """A word-level tokenizer of `vocab` entries (w0, w1, ... + specials) saved the Hugging Face way
    (tokenizer.json + tokenizer_config.json), optionally with a minimal chat template."""
    from tokenizers import Tokenizer, models, pre_tokenizers
    from transformers import PreTrainedTokenizerFast
    specials = ["<unk>", "<s>", "</s>", "<|system|>", "<|user|>", "<|assistant|>"]
    words = specials + [f"w{i}" for i in range(vocab - len(specials))]
    tok = Tokenizer(models.WordLevel({w: i for i, w in enumerate(words)}, unk_token="<unk>"))
    tok.pre_tokenizer = pre_tokenizers.WhitespaceSplit()
    fast = PreTrainedTokenizerFast(tokenizer_object=tok, unk_token="<unk>", bos_token="<s>", eos_token="</s>",
                                   pad_token="</s>")
    if chat_template:
        fast.chat_template = ("{% for m in messages %}<|{{ m['role'] }}|> {{ m['content'] }} {% endfor %}"
                              "{% if add_generation_prompt %}<|assistant|> {% endif %}")
    fast.save_pretrained(path)
    return fast
