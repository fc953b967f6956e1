#!/usr/bin/env python3
"""Per-kernel means of rocprofv3 --pmc counters: every *counter_collection.csv under <dir>,
grouped by (shortened) kernel name; optional substring filter.
    python tools/pmc_kernels.py gpurun_out/xyz out.json [substring]"""
import collections, csv, glob, json, os, re, sys


def short(k):
    k = k.replace("void ", "")
    m = re.match(r"([\w:]+(<[^(]*>)?)", k)
    return (m.group(1) if m else k)[:120]


def main():
    src, dst = sys.argv[1], sys.argv[2]
    flt = sys.argv[3] if len(sys.argv) > 3 else ""
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
    for f in sorted(glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if flt and flt not in k:
                continue
            acc[k][r["Counter_Name"]][(f, r["Dispatch_Id"])] += float(r["Counter_Value"])
    out = {}
    for k, cs in acc.items():
        out[k] = {c: sum(d.values()) / len(d) for c, d in cs.items()}
        out[k]["dispatches"] = max(len(d) for d in cs.values())
        o = out[k]
        if "SQ_VALU_MFMA_BUSY_CYCLES" in o and "GRBM_GUI_ACTIVE" in o:
            # MFMA pipe busy fraction: busy cycles summed over the SIMDs / (kernel cycles x 1024 SIMDs);
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs
            o["mfma_busy_frac"] = o["SQ_VALU_MFMA_BUSY_CYCLES"] / (o["GRBM_GUI_ACTIVE"] / 8 * 1024)
        if "SQ_LDS_BANK_CONFLICT" in o and "SQ_LDS_IDX_ACTIVE" in o and o["SQ_LDS_IDX_ACTIVE"]:
            o["lds_conflict_frac"] = o["SQ_LDS_BANK_CONFLICT"] / o["SQ_LDS_IDX_ACTIVE"]
        if "SQ_LDS_IDX_ACTIVE" in o and "GRBM_GUI_ACTIVE" in o:
            o["lds_busy_frac"] = o["SQ_LDS_IDX_ACTIVE"] / (o["GRBM_GUI_ACTIVE"] / 8 * 256)
    json.dump(out, open(dst, "w"), indent=1)
    for k, o in out.items():
        print(k, {c: (round(v, 3) if v < 10 else round(v)) for c, v in o.items()})


if __name__ == "__main__":
    main()
