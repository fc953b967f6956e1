#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r3sl; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P="python $R/tools/sliced_bench.py --shapes 8192,8192"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o sb -- $P > /dev/null 2>&1
grep -E "sliced|gather" $OUT/stats/sb_kernel_stats.csv | cut -c1-200
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc -o sb -- $P > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE WRITE_SIZE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SALU --output-format csv -d $OUT/pmc2 -o sb -- $P > /dev/null 2>&1
cd $R; rm -f $OUT/*/sb_kernel_trace.csv $OUT/*/sb_agent_info.csv
python tools/pmc_kernels.py $OUT $OUT/sliced_pmc_kernels.json sliced | cut -c1-700
