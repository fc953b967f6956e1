#!/usr/bin/env python3
"""Condense the rocprofv3 --pmc passes of tools/gpu_profile_round.sh into one JSON per bench
configuration: per-launch means of every counter for the GEMV kernel, plus the corrected HBM
traffic bench.py reports as roofline.traffic.

    python tools/pmc_summary.py gpurun_out/round profiles/r01/bench_h8192_single_pmc_summary.json
"""
import collections
import csv
import glob
import json
import os
import sys


def main():
    src, dst = sys.argv[1], sys.argv[2]
    per = collections.defaultdict(lambda: collections.defaultdict(float))   # counter -> dispatch -> sum
    names = collections.Counter()
    for f in sorted(glob.glob(os.path.join(src, "pmc_*", "*counter_collection.csv"))):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "vptq::gemv_" not in k:
                continue
            names[k.split("(")[0].replace("void ", "")] += 1
            per[r["Counter_Name"]][(f, r["Dispatch_Id"])] += float(r["Counter_Value"])
    out = {}
    for c, d in sorted(per.items()):
        v = list(d.values())
        out[c] = {"per_launch_mean": sum(v) / len(v), "launches": len(v)}
    kernel = names.most_common(1)[0][0] if names else "?"
    out["_kernel"] = kernel
    out["_note"] = (
        "rocprofv3 --kernel-trace --pmc <counters> -- python bench.py --steps 5 --warmup 2 "
        "--no-cpu-baseline (one --pmc pass per counter group, tools/gpu_profile_round.sh); "
        "per-launch means over the dispatches of the GEMV kernel, one launch = one layer.  "
        "FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE under-reports a wide coalesced "
        "stream by 2x (MI355X_MICROARCH.md, HBM section): corrected HBM bytes = "
        "2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024.")
    if "FETCH_SIZE" in out:
        out["hbm_bytes_corrected"] = (2 * out["FETCH_SIZE"]["per_launch_mean"] +
                                      out.get("WRITE_SIZE", {"per_launch_mean": 0})["per_launch_mean"]) * 1024
    json.dump(out, open(dst, "w"), indent=1)
    print(dst, kernel, {k: round(v["per_launch_mean"], 1) for k, v in out.items() if isinstance(v, dict)})


if __name__ == "__main__":
    main()
