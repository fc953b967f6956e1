"""`python -m vptq_amd` (reference: vptq/__main__.py)."""
from vptq_amd.app_utils import main

main()
