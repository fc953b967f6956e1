#!/bin/bash
# round 3, last GPU call: GPU suite + smoke + the plain bench line of the final build
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r3last; mkdir -p $OUT
cd $R
timeout 110 python -m pytest tests -q -m gpu -x 2>&1 | tail -3 | tee $OUT/gpu_suite.txt
timeout 40 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $OUT/smoke.txt
timeout 100 python bench.py 2> $OUT/bench_stderr.txt | tail -1 > $OUT/bench_h8192_chain_last.json
cut -c1-300 $OUT/bench_h8192_chain_last.json
