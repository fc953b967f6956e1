from vptq_amd.ops.quant_gemm import (dequant, fused_gemm_max_tokens, quant_gemm, quant_gemm_flags, quant_gemm_fused,
                                     quant_gemv_v2)

__all__ = ["dequant", "quant_gemm", "quant_gemv_v2"]
