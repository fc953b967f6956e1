#!/bin/bash
# round 4, step 38: final build: the GPU suite, smoke, the sliced fuzzers
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4s38; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --tb=short 2>&1 | tail -8 | tee $OUT/suite.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-1500 | tee $OUT/smoke.txt
timeout 400 python tools/gpu_fuzz.py --sliced --cases 60 --seed 1901 2>&1 | tail -1 | tee $OUT/fuzz_sliced.txt
timeout 300 python tools/gpu_fuzz.py --sliced --cases 30 --seed 1902 --dtype bf16 2>&1 | tail -1 | tee -a $OUT/fuzz_sliced.txt
