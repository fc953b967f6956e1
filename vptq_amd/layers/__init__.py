from vptq_amd.layers.vqlinear import VQuantLinear, chain_prefetch
from vptq_amd.layers.model_base import AutoModelForCausalLM

__all__ = ["VQuantLinear", "chain_prefetch", "AutoModelForCausalLM"]
