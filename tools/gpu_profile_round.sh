#!/bin/bash
# Produce the round's bench lines + rocprofv3 evidence under gpurun_out/round/.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/round; mkdir -p $OUT
cd $R
timeout 600 python bench.py 2>&1 | tail -1 > $OUT/bench_h8192_single.json
timeout 600 python bench.py --mode grouped --no-cpu-baseline 2>&1 | tail -1 > $OUT/bench_h8192_grouped.json
timeout 600 python bench.py --hidden 4096 2>&1 | tail -1 > $OUT/bench_h4096_single.json
timeout 600 python bench.py --hidden 4096 --mode grouped --no-cpu-baseline 2>&1 | tail -1 > $OUT/bench_h4096_grouped.json
timeout 600 python bench.py --exact --no-cpu-baseline 2>&1 | tail -1 > $OUT/bench_h8192_single_exact.json
timeout 600 python bench.py --exact --mode grouped --no-cpu-baseline 2>&1 | tail -1 > $OUT/bench_h8192_grouped_exact.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_sq -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq2 -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
cd $R
# kernel A/B (VALU vs persistent MFMA, default vs exact arithmetic) and per-projection shapes
for H in 8192 4096; do timeout 300 python tools/microbench.py --hidden $H --group 4 --out $OUT/kernel_ab_h$H.json > /dev/null 2>&1; done
for M in 70b 8b; do timeout 300 python tools/shape_bench.py --model $M --out $OUT/shapes_llama3_$M.json > /dev/null 2>&1; done
rm -f $OUT/*/bench_kernel_trace.csv $OUT/*/bench_agent_info.csv   # large / uninteresting
for f in $OUT/bench_*.json; do echo $f; cut -c1-200 $f; done
cut -c1-160 $OUT/stats/bench_kernel_stats.csv | head -3
