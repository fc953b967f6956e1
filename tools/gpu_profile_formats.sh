#!/bin/bash
# rocprofv3 kernel stats of the gather / gatherx / dequant kernels (copy the CSVs into profiles/rNN/)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4c; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/formats -o k -- python $R/tools/format_bench.py --formats v8-k65536-256,v16-k65536-65536,v16-k65536-1024,v12-k65536-4096 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prefill -o k -- python $R/tools/prefill_bench.py --tokens 2048 --shapes "8192,8192" --dtypes f16 > /dev/null 2>&1
cd $R
rm -f $OUT/*/k_kernel_trace.csv $OUT/*/k_agent_info.csv
for f in $OUT/formats/k_kernel_stats.csv $OUT/prefill/k_kernel_stats.csv; do echo $f; cut -c1-200 $f | grep -v "at::native\|Cijk\|hipblas" | head -12; done
