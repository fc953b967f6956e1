#!/bin/bash
# round 4, step 33: the final build: GPU suite, smoke, the sliced family fuzzer with random token counts (more cases)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4s33; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --tb=short 2>&1 | tail -8 | tee $OUT/suite.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | cut -c1-1500 | tee $OUT/smoke.txt
timeout 600 python tools/gpu_fuzz.py --sliced --cases 150 --seed 1801 2>&1 | tail -2 | tee $OUT/fuzz_sliced.txt
timeout 400 python tools/gpu_fuzz.py --sliced --cases 80 --seed 1802 --dtype bf16 2>&1 | tail -2 | tee -a $OUT/fuzz_sliced.txt
timeout 300 python tools/gpu_fuzz.py --formats --cases 100 --seed 1803 2>&1 | tail -1 | tee $OUT/fuzz_formats.txt
timeout 300 python tools/gpu_fuzz.py --tokens --cases 30 --seed 1804 2>&1 | tail -1 | tee $OUT/fuzz_tokens.txt
