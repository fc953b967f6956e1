#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prefill; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o pf -- python $R/tools/prefill_bench.py > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc -o pf -- python $R/tools/prefill_bench.py > /dev/null 2>&1
cd $R; rm -f $OUT/*/pf_kernel_trace.csv $OUT/*/pf_agent_info.csv
cut -c1-150 $OUT/stats/pf_kernel_stats.csv | head -6
ls $OUT/pmc
