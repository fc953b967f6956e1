"""Import microsoft/VPTQ read-only from /root/reference (build container only).

Two shims: (1) stub ``sentence_transformers`` which
vptq/utils/pack.py:14 imports at module top level; (2) make
``importlib.metadata.version("vptq")`` (vptq/__init__.py:9) succeed although
the package is not pip-installed.  ``vptq.libvptq`` is absent, so the
reference runs its pure-torch CPU path (vptq/ops/quant_gemm.py:28-40,247-274).
"""
import importlib.metadata as _md
import os
import sys
import types

REFERENCE_PATH = "/root/reference"


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_PATH, "vptq"))


def load_reference(path: str = REFERENCE_PATH):
    if "vptq" in sys.modules and getattr(sys.modules["vptq"], "__file__", "").startswith(path):
        return sys.modules["vptq"]
    st = types.ModuleType("sentence_transformers")
    st2 = types.ModuleType("sentence_transformers.SentenceTransformer")

    class SentenceTransformer:  # noqa: D401 - stub
        pass

    st2.SentenceTransformer = SentenceTransformer
    st.SentenceTransformer = st2
    sys.modules.setdefault("sentence_transformers", st)
    sys.modules.setdefault("sentence_transformers.SentenceTransformer", st2)
    orig = _md.version
    _md.version = lambda n: "0.0.5.post1" if n == "vptq" else orig(n)
    # the repo root also has a `vptq` alias package: make sure the reference wins
    saved = sys.modules.pop("vptq", None)
    sys.path.insert(0, path)
    try:
        import vptq  # noqa: E402
    finally:
        sys.path.remove(path)
        _md.version = orig
    assert vptq.__file__.startswith(path), vptq.__file__
    del saved
    return vptq
