#!/bin/bash
set -x
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout 600 python bench.py 2>&1 | tail -1 > gpurun_out/bench_8192_single.json; cat gpurun_out/bench_8192_single.json
timeout 600 python bench.py --mode grouped --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_8192_grouped.json; cat gpurun_out/bench_8192_grouped.json
timeout 600 python bench.py --hidden 4096 2>&1 | tail -1 > gpurun_out/bench_4096_single.json; cat gpurun_out/bench_4096_single.json
timeout 600 python bench.py --fast-math --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_8192_single_fast.json; cat gpurun_out/bench_8192_single_fast.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $R/gpurun_out/pmc_sq -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_sq2 -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
cd $R
ls -R gpurun_out | head -50
cut -c1-150 gpurun_out/prof_bench/bench_kernel_stats.csv | head -6
rm -f gpurun_out/*/*kernel_trace.csv.bak
