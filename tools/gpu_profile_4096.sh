#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/round4096; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python $R/bench.py --hidden 4096 --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o bench -- python $R/bench.py --hidden 4096 --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o bench -- python $R/bench.py --hidden 4096 --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
cd $R; rm -f $OUT/*/bench_kernel_trace.csv $OUT/*/bench_agent_info.csv
cut -c1-160 $OUT/stats/bench_kernel_stats.csv | head -3
