#!/bin/bash
# PMC evidence for gemv_gatherx (v16-k65536-65536) and the T = 24 gather kernel with its LDS residual table
R=$GRAFT_REPO_ROOT; OUT=gpurun_out/r4e; mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
for f in v16-k65536-65536 v8-k65536-256; do
CMD="python $R/tools/format_bench.py --formats $f"
timeout 300 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --output-format csv -d $R/$OUT/$f/pmc_a -o g -- $CMD > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $R/$OUT/$f/pmc_c -o g -- $CMD > /dev/null 2>&1
done
cd $R; rm -f $OUT/*/*/g_kernel_trace.csv $OUT/*/*/g_agent_info.csv
python tools/pmc_kernels.py $OUT/v16-k65536-65536 $OUT/gatherx_v16_pmc_summary.json gemv_gatherx
python tools/pmc_kernels.py $OUT/v8-k65536-256 $OUT/gather_t24_pmc_summary.json gemv_gather_kernel
