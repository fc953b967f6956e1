"""Import alias: `import vptq` resolves to vptq_amd so that Hugging Face
Transformers' VPTQ integration (`from vptq import VQuantLinear`,
transformers/integrations/vptq.py) and code written against microsoft/VPTQ run
unchanged on MI355X."""
import sys

import vptq_amd
from vptq_amd import AutoModelForCausalLM, VQuantLinear, __version__, ops  # noqa: F401
from vptq_amd import layers, utils  # noqa: F401

sys.modules[__name__ + ".ops"] = ops
sys.modules[__name__ + ".layers"] = layers
sys.modules[__name__ + ".utils"] = utils

__all__ = ["AutoModelForCausalLM", "VQuantLinear", "ops", "__version__"]
