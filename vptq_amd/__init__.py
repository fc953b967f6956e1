"""vptq_amd — MI355X-native fused dequant+GEMV for VPTQ-quantised linears.

Keeps the `vptq.ops` / `vptq.VQuantLinear` API of microsoft/VPTQ
(vptq/__init__.py:7-14) for the VQuantLinear.forward hot path; compute runs in
hand-written HIP kernels (vptq_amd/csrc) behind the C ABI in include/vptq_hip.h.
"""
__version__ = "0.0.5.post1"

from vptq_amd import ops  # noqa: E402
from vptq_amd.layers import AutoModelForCausalLM, VQuantLinear  # noqa: E402
from vptq_amd._backend import arithmetic, set_arithmetic  # noqa: E402  ("reference" by default; "selective" / "folded" = the opt-in fast forms)

__all__ = ["AutoModelForCausalLM", "VQuantLinear", "ops", "arithmetic", "set_arithmetic", "__version__"]
