from vptq_amd.layers.vqlinear import VQuantLinear, chain_prefetch

__all__ = ["VQuantLinear", "chain_prefetch"]
