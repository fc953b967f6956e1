#!/bin/bash
set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
for H in 8192 4096; do
  timeout 300 python tools/microbench.py --hidden $H --out gpurun_out/mb2_${H}.json 2>&1 | grep -E "^(exact|fast|generic|Traceback|Assert)" 
done
timeout 300 python tools/microbench.py --hidden 8192 --group 4 --out gpurun_out/mb2_8192_g4.json 2>&1 | grep -E "^(exact|fast)"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof2 -o mb -- python $GRAFT_REPO_ROOT/tools/microbench.py --hidden 8192 --iters 3 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/prof2 -name "*kernel_stats*" | head -1 | xargs -I{} head -12 {}
