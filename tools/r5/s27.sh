#!/bin/bash
# round 5, session 27: the last build (window parts in) - GPU suite, smoke, sliced fuzz, counting fuzz through the module
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r5s27; mkdir -p $OUT
cd $R
export PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider --tb=short 2>&1 | tail -40 > $OUT/gpu_suite.txt
grep -E "^FAILED|passed|failed" $OUT/gpu_suite.txt | tail -12
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | cut -c1-200 | tee $OUT/smoke.txt
timeout 400 python tools/gpu_fuzz.py --sliced --cases 36 --seed 171 2>&1 | grep -v amdgpu.ids > $OUT/fuzz_sliced_f16.txt; tail -1 $OUT/fuzz_sliced_f16.txt
timeout 300 python tools/gpu_fuzz.py --sliced --cases 16 --seed 172 --dtype bf16 2>&1 | grep -v amdgpu.ids > $OUT/fuzz_sliced_bf16.txt; tail -1 $OUT/fuzz_sliced_bf16.txt
timeout 300 python tools/gpu_sliced_count.py --layers 1600 --seed 5 2>&1 | grep -v amdgpu.ids | grep -v "layers after" > $OUT/sliced_count_f16.txt; tail -3 $OUT/sliced_count_f16.txt
timeout 300 python tools/gpu_sliced_count.py --layers 800 --seed 6 --dtype bf16 2>&1 | grep -v amdgpu.ids | grep -v "layers after" > $OUT/sliced_count_bf16.txt; tail -1 $OUT/sliced_count_bf16.txt
