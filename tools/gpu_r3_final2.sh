#!/bin/bash
# round 3, last session: GPU suite + smoke + the bench line with its rocprofv3 stats / PMC passes (the chain kernel's source
# gained compile-time knobs and bench.py the power soak: the default build is re-verified and the evidence refreshed)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r3fin2; mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -5 | tee $OUT/gpu_suite.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $OUT/smoke.txt
bash tools/gpu_r3_bench1.sh > $OUT/bench1.log 2>&1; tail -3 $OUT/bench1.log | cut -c1-300
