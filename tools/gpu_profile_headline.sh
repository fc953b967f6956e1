#!/bin/bash
# rocprofv3 kernel stats of the default bench command (no extras / CPU baseline: the same timed workload)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r5s; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/headline -o k -- python $R/bench.py --no-extras --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
cd $R; rm -f $OUT/headline/k_kernel_trace.csv $OUT/headline/k_agent_info.csv
timeout 300 python bench.py --no-extras --no-cpu-baseline > $OUT/bench_plain.json 2>/dev/null
head -3 $OUT/headline/k_kernel_stats.csv | cut -c1-220
python -c "
import json
for f in ('bench_under_rocprof', 'bench_plain'):
    d = json.load(open('gpurun_out/r5s/%s.json' % f)); print(f, d['value'], d['roofline']['us_per_launch'])"
