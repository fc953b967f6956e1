#!/bin/bash
# rocprofv3 evidence for the fused dequant + GEMM kernel (and the dense route beside it)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/fused; mkdir -p $OUT
CMD="python $R/tools/prefill_bench.py --tokens ${TOKENS:-8192} --shapes ${SHAPES:-8192,8192} --dtypes ${DTYPES:-f16,bf16}"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o pf -- $CMD > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_a -o pf -- $CMD > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_b -o pf -- $CMD > /dev/null 2>&1
cd $R; rm -f $OUT/*/pf_kernel_trace.csv $OUT/*/pf_agent_info.csv
cut -c1-160 $OUT/stats/pf_kernel_stats.csv | head -8
python tools/pmc_kernels.py $OUT $OUT/pmc_summary.json
