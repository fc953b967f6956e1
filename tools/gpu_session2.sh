#!/bin/bash
set -x
mkdir -p gpurun_out
timeout 600 tools/_build/ubench > gpurun_out/ubench.txt 2>&1
cat gpurun_out/ubench.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof1 -o mb -- python $GRAFT_REPO_ROOT/tools/microbench.py --hidden 8192 --iters 3 2>&1 | tail -4
cd $GRAFT_REPO_ROOT
ls -R gpurun_out/prof1 | head -20
find gpurun_out/prof1 -name "*kernel_stats*" | head -1 | xargs -I{} head -20 {}
