#!/bin/bash
# round 4, GPU call 15: full suite + smoke after the sliced-family / token-route changes
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4s15; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -15 | tee $OUT/gpu_suite.txt
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke.txt
