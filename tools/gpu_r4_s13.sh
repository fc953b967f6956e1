#!/bin/bash
# round 4, GPU call 13: fuzz of the sliced family (random v / k / kr / shapes / skewed indices)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4s13; mkdir -p $OUT
cd $R
timeout 900 python tools/gpu_fuzz.py --sliced --cases 80 --seed 21 2>&1 | grep -v amdgpu.ids | tail -84 | tee $OUT/fuzz_sliced_family_f16.txt
timeout 600 python tools/gpu_fuzz.py --sliced --cases 40 --seed 22 --dtype bf16 2>&1 | grep -v amdgpu.ids | tail -44 | tee $OUT/fuzz_sliced_family_bf16.txt
