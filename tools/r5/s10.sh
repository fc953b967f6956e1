#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r5s10; mkdir -p $OUT
cd $R
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gemv_sliced_gpu.py -m gpu -q -p no:cacheprovider --tb=short -x 2>&1 | tail -6
timeout 600 python tools/gpu_fuzz.py --sliced --cases 60 --seed 51 2>&1 | grep -v amdgpu.ids | grep -v "^  " > $OUT/fuzz_sliced_f16.txt; tail -3 $OUT/fuzz_sliced_f16.txt
timeout 400 python tools/gpu_fuzz.py --sliced --cases 30 --seed 52 --dtype bf16 2>&1 | grep -v amdgpu.ids | grep -v "^  " > $OUT/fuzz_sliced_bf16.txt; tail -3 $OUT/fuzz_sliced_bf16.txt
timeout 400 python tools/gpu_fuzz.py --sliced --cases 40 --seed 53 2>&1 | grep -v amdgpu.ids | grep -v "^  " > $OUT/fuzz_sliced_f16_b.txt; tail -2 $OUT/fuzz_sliced_f16_b.txt
rm -f gpucore.*
