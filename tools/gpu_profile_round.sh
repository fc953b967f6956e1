#!/bin/bash
# Produce the round's bench lines + rocprofv3 evidence under gpurun_out/round/ (copy into profiles/rNN/).
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/round; mkdir -p $OUT
cd $R
timeout 900 python bench.py 2>&1 | tail -1 > $OUT/bench_h8192_single.json
timeout 600 python bench.py --mode grouped --no-cpu-baseline --no-extras 2>&1 | tail -1 > $OUT/bench_h8192_grouped.json
timeout 600 python bench.py --hidden 4096 --no-extras 2>&1 | tail -1 > $OUT/bench_h4096_single.json
timeout 600 python bench.py --exact --no-cpu-baseline --no-extras 2>&1 | tail -1 > $OUT/bench_h8192_single_exact.json
timeout 600 python bench.py --mode tp_row --no-extras 2>&1 | tail -1 > $OUT/bench_tp_row_n1.json
B="python $R/bench.py --no-cpu-baseline --no-extras --regions 1"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- $B --steps 20 --warmup 5 > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o bench -- $B --steps 5 --warmup 2 > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o bench -- $B --steps 5 --warmup 2 > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_sq -o bench -- $B --steps 5 --warmup 2 > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq2 -o bench -- $B --steps 5 --warmup 2 > /dev/null 2>&1
# the other kernels of the round: stats + LDS / wait counters
for what in "lds:python $R/tools/microbench.py --hidden 8192 --k 8192 --kr 256 --variants default --no-copy --iters 3" \
            "tok16:python $R/tools/tokens_bench.py --shapes 8192,8192 --tokens 16"; do
  n=${what%%:*}; c=${what#*:}
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/k_$n/stats -o k -- $c > /dev/null 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/k_$n/pmc_a -o k -- $c > /dev/null 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/k_$n/pmc_b -o k -- $c > /dev/null 2>&1
done
cd $R
rm -f $OUT/*/bench_kernel_trace.csv $OUT/*/bench_agent_info.csv $OUT/k_*/*/k_kernel_trace.csv $OUT/k_*/*/k_agent_info.csv
python tools/pmc_summary.py $OUT $OUT/bench_h8192_single_pmc_summary.json
python tools/pmc_kernels.py $OUT/k_lds $OUT/gemv_lds_pmc_summary.json gemv_lds > /dev/null
python tools/pmc_kernels.py $OUT/k_tok16 $OUT/gemm_k256_pmc_summary.json gemm_k256 > /dev/null
for f in $OUT/bench_*.json; do echo $f; cut -c1-260 $f; done
cut -c1-170 $OUT/stats/bench_kernel_stats.csv | head -3
cut -c1-170 $OUT/k_lds/stats/k_kernel_stats.csv | head -3
cut -c1-170 $OUT/k_tok16/stats/k_kernel_stats.csv | head -3
cat $OUT/gemv_lds_pmc_summary.json $OUT/gemm_k256_pmc_summary.json | head -60
