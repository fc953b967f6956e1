#!/bin/bash
# round 5, session 21: rocprofv3 kernel stats + PMC passes of the last kernels of the round: two tokens in one pass of the exact sliced
# kernel (gemv_sliced<EX, TOK = 2>, 8192^2 v8-k65536-256) and a 28672 x 8192 layer as two column parts
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r5s21; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in "tok2_r256:tools/sliced_tokens_exact_bench.py --kr 256 --tokens 2 --shapes 8192,8192" "tok3_r256:tools/sliced_tokens_exact_bench.py --kr 256 --tokens 3 --shapes 8192,8192" "parts_r256:tools/sliced_bench.py --exact --kr 256 --shapes 28672,8192 --ring 4"; do
  n=${cfg%%:*}; a=${cfg#*:}
  C="python $R/$a"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$n/stats -o k -- $C > /dev/null 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/$n/pmc_a -o k -- $C > /dev/null 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/$n/pmc_b -o k -- $C > /dev/null 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/$n/pmc_c -o k -- $C > /dev/null 2>&1
  find $OUT/$n -name "k_kernel_trace.csv" -delete; find $OUT/$n -name "k_agent_info.csv" -delete
  python $R/tools/pmc_kernels.py $OUT/$n $OUT/sliced_${n}_pmc_summary.json gemv_sliced | cut -c1-1200
  f=$(find $OUT/$n/stats -name "*kernel_stats.csv" | head -1); grep "gemv_sliced\|gemv_gather" $f | cut -c1-260 > $OUT/sliced_${n}_kernel_stats.csv; cat $OUT/sliced_${n}_kernel_stats.csv
done
