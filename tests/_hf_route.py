"""Hugging Face's own VPTQ loading route (transformers.AutoModelForCausalLM.from_pretrained ->
VptqHfQuantizer -> replace_with_vptq_linear -> `from vptq import VQuantLinear`), exercised on a
synthetic checkpoint; SURVEY.md section 3.3 / 8(f) rank 2.

transformers 5.15.0 has a bug on this route: `replace_with_vptq_linear` indexes
`model._modules[module_name]` with the DOTTED name of a nested module
(transformers/integrations/vptq.py, the line `model._modules[module_name].requires_grad_(False)`),
which raises KeyError for every real model.  `fixed_replace_with_vptq_linear` below is that
function with that one line resolved through `get_submodule`; `hf_from_pretrained` first tries
the unmodified route and only on that KeyError installs the fixed function - everything else
(quantizer hooks, `from vptq import VQuantLinear`, meta construction, HF's weight loading into
the module's parameters by state-dict name / dtype) is Hugging Face's own code.
"""
import contextlib
import traceback


def fixed_replace_with_vptq_linear(model, modules_to_not_convert=None, quantization_config=None):
    import torch
    import torch.nn as nn
    from transformers.quantizers.quantizers_utils import should_convert_module
    from vptq import VQuantLinear
    shared, per_layer = quantization_config.shared_layer_config, quantization_config.config_for_layers
    for module_name, module in list(model.named_modules()):
        if not should_convert_module(module_name, modules_to_not_convert) or not isinstance(module, nn.Linear):
            continue
        lp = per_layer.get(module_name, None) or shared.get(module_name.rsplit(".")[1], None)
        with torch.device("meta"):
            new = VQuantLinear(
                module.in_features, module.out_features, vector_lens=lp["vector_lens"],
                num_centroids=lp["num_centroids"], num_res_centroids=lp["num_res_centroids"],
                group_num=lp["group_num"], group_size=lp["group_size"], outlier_size=lp["outlier_size"],
                indices_as_float=lp["indices_as_float"], enable_norm=lp["enable_norm"],
                enable_perm=lp["enable_perm"], is_indice_packed=True, enable_proxy_error=False,
                bias=module.bias is not None)
        model.get_submodule(module_name).requires_grad_(False)     # <- the upstream line, fixed
        model.set_submodule(module_name, new)
    return model


@contextlib.contextmanager
def patched_integration():
    import transformers.integrations as integ
    import transformers.integrations.vptq as iv
    old_a, old_b = integ.replace_with_vptq_linear, iv.replace_with_vptq_linear
    integ.replace_with_vptq_linear = iv.replace_with_vptq_linear = fixed_replace_with_vptq_linear
    try:
        yield
    finally:
        integ.replace_with_vptq_linear, iv.replace_with_vptq_linear = old_a, old_b


def hf_from_pretrained(path, **kw):
    """-> (model, upstream_bug: bool)"""
    import transformers
    try:
        return transformers.AutoModelForCausalLM.from_pretrained(path, **kw), False
    except KeyError:
        tb = traceback.format_exc()
        if "integrations/vptq.py" not in tb:
            raise
    with patched_integration():
        return transformers.AutoModelForCausalLM.from_pretrained(path, **kw), True
