#!/bin/bash
# round 5, session 11: rocprofv3 kernel stats + PMC passes of the new kernels (gemv_sliced<EX>, <EX, RG>, folded) on 8192^2 layers
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r5s11; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in "ex_r256:--exact --kr 256" "ex_r0:--exact --kr 0" "rg_r65536:--exact --kr 65536" "folded_r256:--kr 256"; do
  n=${cfg%%:*}; a=${cfg#*:}
  C="python $R/tools/sliced_bench.py $a --shapes 8192,8192 --ring 8"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$n/stats -o k -- $C > /dev/null 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/$n/pmc_a -o k -- $C > /dev/null 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/$n/pmc_b -o k -- $C > /dev/null 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/$n/pmc_c -o k -- $C > /dev/null 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/$n/pmc_d -o k -- $C > /dev/null 2>&1
  find $OUT/$n -name "k_kernel_trace.csv" -delete; find $OUT/$n -name "k_agent_info.csv" -delete
  python $R/tools/pmc_kernels.py $OUT/$n $OUT/sliced_${n}_pmc_summary.json gemv_sliced | cut -c1-900
  f=$(find $OUT/$n/stats -name "*kernel_stats.csv" | head -1); grep "gemv_sliced\|gemv_gather" $f | cut -c1-220 > $OUT/sliced_${n}_kernel_stats.csv; cat $OUT/sliced_${n}_kernel_stats.csv
done
