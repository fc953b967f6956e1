#!/bin/bash
# PMC breakdown of the persistent MFMA kernel in steady state (grouped x4 bench)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/grouped; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/pmc_a -o bench -- python $R/bench.py --mode grouped --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_b -o bench -- python $R/bench.py --mode grouped --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
cd $R; rm -f $OUT/*/bench_kernel_trace.csv $OUT/*/bench_agent_info.csv
python tools/pmc_summary.py $OUT $OUT/summary.json
