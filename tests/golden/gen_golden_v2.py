#!/usr/bin/env python3
"""Golden vectors for the v2 wire format, produced by the REFERENCE'S OWN TEST FILE.

    python tests/golden/gen_golden_v2.py         # writes tests/golden/v2/*.npz

Imports /root/reference/tests/test_quant_gemv.py (read-only; `vptq` = the reference through
_refshim.py), builds the inputs with its `create_test_data` (:301-361: seed 1234,
normal(0.02, 0.5), cyclic indices) on the CPU and records the result of its `ground_truth`
(:49-109).  The reference's two test configurations (bf16, the dtype its test runs) plus an
fp16 one and a biased, two-token one.  The index tensors are the reference's cyclic
`arange(k)` pattern and are rebuilt by the loader; everything else is stored.
"""
import importlib.util
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from _refshim import load_reference, REFERENCE_PATH  # noqa: E402
import gen_golden as gg  # noqa: E402

V2_CASES = [
    # the reference's own two configurations (tests/test_quant_gemv.py:196-219)
    dict(name="ref_residual_indices_uint8", in_features=1024, out_features=2048, num_centroids=8192,
         num_res_centroids=256, dtype="bf16"),
    dict(name="ref_residual_indices_uint16", in_features=1024, out_features=1024, num_centroids=8192,
         num_res_centroids=512, dtype="bf16"),
    dict(name="f16_k4096_r256", in_features=512, out_features=1024, num_centroids=4096,
         num_res_centroids=256, dtype="f16"),
    dict(name="f16_k8192_r512_t2_bias", in_features=1024, out_features=512, num_centroids=8192,
         num_res_centroids=512, dtype="f16", length=2, with_bias=True),
]


def load_reference_test_module():
    """The reference's test file as a module (its `import vptq` gets the reference package)."""
    load_reference()
    spec = importlib.util.spec_from_file_location(
        "ref_test_quant_gemv", os.path.join(REFERENCE_PATH, "tests", "test_quant_gemv.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    rt = load_reference_test_module()
    os.makedirs(os.path.join(HERE, "v2"), exist_ok=True)
    for c in V2_CASES:
        cfg = dict(c, vector_length=8, device=torch.device("cpu"), dtype=gg.TORCH_DT[c["dtype"]])
        d = rt.create_test_data(cfg)
        if c.get("with_bias"):
            d["bias"] = rt._create_tensor((1, c["out_features"]), 0.02, 0.5, cfg["dtype"], cfg["device"])
        y = rt.ground_truth(x=d["x"], bias=d["bias"], indices=d["indices"], centroids=d["centroids"],
                            res_indices=d["residual_indices"], res_centroids=d["residual_centroids"],
                            scale_weights=d["scale_weights"], scale_bias=d["scale_bias"],
                            vector_len=d["vector_len"], out_features=d["out_features"])
        meta = dict(name=c["name"], in_features=c["in_features"], out_features=c["out_features"],
                    num_centroids=c["num_centroids"], num_res_centroids=c["num_res_centroids"],
                    vector_len=8, dtype=c["dtype"], length=c.get("length", 1),
                    res_index_dtype=str(d["residual_indices"].dtype).replace("torch.", ""),
                    index_dtype=str(d["indices"].dtype).replace("torch.", ""),
                    indices="arange(k).repeat(n // k)  (tests/test_quant_gemv.py:21-31)",
                    torch=torch.__version__)
        arrs = dict(config=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8),
                    x=gg.bits(d["x"]), centroids=gg.bits(d["centroids"]),
                    res_centroids=gg.bits(d["residual_centroids"]),
                    scale_weights=gg.bits(d["scale_weights"]), scale_bias=gg.bits(d["scale_bias"]),
                    y=gg.bits(y))
        if d["bias"] is not None:
            arrs["bias"] = gg.bits(d["bias"])
        out = os.path.join(HERE, "v2", c["name"] + ".npz")
        np.savez_compressed(out, **arrs)
        print(f"{c['name']}: y{tuple(y.shape)} -> {os.path.getsize(out) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
