#!/bin/bash
# round 4, step 31: the GPU suite, the sliced fuzzers with random token counts, smoke
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4s31; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --tb=short 2>&1 | tail -25 | tee $OUT/suite.txt
timeout 400 python tools/gpu_fuzz.py --sliced --cases 80 --seed 1701 2>&1 | tail -3 | tee $OUT/fuzz_sliced.txt
timeout 300 python tools/gpu_fuzz.py --sliced --cases 40 --seed 1702 --dtype bf16 2>&1 | tail -2 | tee -a $OUT/fuzz_sliced.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee $OUT/smoke.txt
