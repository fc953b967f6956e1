#!/bin/bash
# PMC evidence for gemv_lds_mfma_kernel (v8-k8192-256 and v8-k4096-0, 8192^2, one token)
R=$GRAFT_REPO_ROOT; OUT=gpurun_out/r5g; mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
for f in v8-k8192-256 v8-k4096-0; do
CMD="python $R/tools/format_bench.py --formats $f"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/$OUT/$f/pmc_a -o g -- $CMD > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $R/$OUT/$f/pmc_b -o g -- $CMD > /dev/null 2>&1
done
cd $R; rm -f $OUT/*/*/g_kernel_trace.csv $OUT/*/*/g_agent_info.csv
python tools/pmc_kernels.py $OUT/v8-k8192-256 $OUT/lds_mfma_k8192_256_pmc_summary.json gemv_lds_mfma
python tools/pmc_kernels.py $OUT/v8-k4096-0 $OUT/lds_mfma_k4096_0_pmc_summary.json gemv_lds_mfma
