#!/bin/bash
# round 3: PMC passes of the chain kernel (ring of 32 layers 8192^2 in one launch): where do the wave-cycles go
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r3pmc; mkdir -p $OUT
C="python $R/tools/chain_bench.py --hidden 8192 --modes chain32 --reps 1 --iters 2"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o k -- $C > /dev/null 2>&1
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH" \
           "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_VALU_MFMA_MOPS_F16"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc_$i -o k -- $C > $OUT/pmc_$i.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections, json, os
out = collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/r3pmc/pmc_*/**/*counter_collection.csv", recursive=True)):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if "gemv_k256t" in r["Kernel_Name"]:
            per[(r["Counter_Name"], r["Dispatch_Id"])] += float(r["Counter_Value"])
    byc = collections.defaultdict(list)
    for (c, d), v in per.items():
        byc[c].append(v)
    for c, v in byc.items():
        out[c] = [sum(v) / len(v), len(v)]
json.dump(out, open("gpurun_out/r3pmc/chain32_pmc_summary.json", "w"), indent=1)
for c, v in sorted(out.items()):
    print(f"{c:36s} {v[0]:16.1f}  ({v[1]} launches)")
PY
grep -h "gemv_k256t" $OUT/stats/*kernel_stats.csv | cut -c1-200
rm -rf $OUT/pmc_*/*/*kernel_trace.csv $OUT/stats/*kernel_trace.csv
