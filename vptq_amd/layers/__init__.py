from vptq_amd.layers.vqlinear import VQuantLinear, SiblingGroup, chain_prefetch, link_siblings
from vptq_amd.layers.model_base import AutoModelForCausalLM

__all__ = ["VQuantLinear", "SiblingGroup", "chain_prefetch", "link_siblings", "AutoModelForCausalLM"]
