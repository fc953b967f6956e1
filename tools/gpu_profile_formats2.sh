#!/bin/bash
# rocprofv3 kernel stats of the kernels added in session 3 (gemv_lds_mfma, gemv_gatherx for the other vector lengths / outlier formats)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r5f; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/formats -o k -- python $R/tools/format_bench.py --formats v8-k8192-256,v8-k4096-0,v6-k4096-0,v4-k256-256,v8-k65536-256-o128 > /dev/null 2>&1
cd $R
rm -f $OUT/*/k_kernel_trace.csv $OUT/*/k_agent_info.csv
cut -c1-220 $OUT/formats/k_kernel_stats.csv | grep -v "at::native\|Cijk\|hipblas" | head -14
