from vptq_amd.layers.vqlinear import VQuantLinear

__all__ = ["VQuantLinear"]
