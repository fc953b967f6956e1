"""`python -m vptq` (reference: vptq/__main__.py) - the command line of this package."""
from vptq_amd.app_utils import main

main()
