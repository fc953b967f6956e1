#!/bin/bash
# round 4, step 39: rocprofv3 evidence for the 2 - 4 token sliced kernel: kernel stats of the timing tool, then counters (own runs)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4s39; mkdir -p $OUT
CMD="python $R/tools/sliced_tokens_bench.py --v 8 --kr 256 --shapes 8192,8192 --only-one-launch"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o tok -- $CMD > $OUT/under_rocprofv3.txt 2> /dev/null
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_sq -o tok -- $CMD > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/pmc_sq2 -o tok -- $CMD > /dev/null 2>&1
cd $R
python - <<'PY'
import csv, os, json, collections
out = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r4s39")
res = {}
for d in ("pmc_sq", "pmc_sq2"):
    f = os.path.join(out, d, "tok_counter_collection.csv")
    if not os.path.exists(f):
        continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if "gemv_sliced_tok_kernel" in r["Kernel_Name"]:
            name = r["Kernel_Name"].split("(")[0].replace("void vptq::", "")
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        res.setdefault(k, {}).update({c: sum(v) / len(v) for c, v in cs.items()})
        res[k]["dispatches_" + d] = len(next(iter(cs.values())))
json.dump(res, open(os.path.join(out, "tok_pmc_summary.json"), "w"), indent=1)
print(json.dumps(res, indent=1)[:3000])
PY
rm -f $OUT/*/tok_kernel_trace.csv $OUT/*/tok_agent_info.csv $OUT/*/tok_counter_collection.csv
grep -h "gemv_sliced_tok" $OUT/stats/tok_kernel_stats.csv | cut -c1-260
cat $OUT/under_rocprofv3.txt | tail -2
