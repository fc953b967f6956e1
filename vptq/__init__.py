"""Import alias: `import vptq` resolves to vptq_amd so that Hugging Face
Transformers' VPTQ integration (`from vptq import VQuantLinear`,
transformers/integrations/vptq.py) and code written against microsoft/VPTQ run
unchanged on MI355X."""
import sys

import vptq_amd
from vptq_amd import AutoModelForCausalLM, VQuantLinear, __version__, ops  # noqa: F401
from vptq_amd import layers, utils  # noqa: F401
from vptq_amd import app_utils  # noqa: F401

import vptq_amd.layers.model_base  # noqa: E402,F401
import vptq_amd.layers.vqlinear  # noqa: E402,F401
import vptq_amd.ops.quant_gemm  # noqa: E402,F401
import vptq_amd.utils.pack  # noqa: E402,F401
import vptq_amd.utils.shard  # noqa: E402,F401
import vptq_amd.app_utils  # noqa: E402,F401

# Every module of the package under its reference name, leaves included: without the leaves
# `from vptq.layers.vqlinear import VQuantLinear` (the reference's canonical import path)
# would execute vqlinear.py a second time and produce a second, unrelated VQuantLinear class
# that `isinstance` checks in link_siblings / absorb_perm / the shard tools do not recognise.
for _name, _mod in list(sys.modules.items()):
    if _name == "vptq_amd" or not _name.startswith("vptq_amd.") or _mod is None:
        continue
    if _name.startswith("vptq_amd._") :
        continue  # private binding module: no reference counterpart
    sys.modules[__name__ + _name[len("vptq_amd"):]] = _mod

__all__ = ["AutoModelForCausalLM", "VQuantLinear", "ops", "__version__"]
