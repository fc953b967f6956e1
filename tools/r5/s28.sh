#!/bin/bash
# round 5, session 28: counting fuzz through the module with EVERY layer the library takes sent over its layouts for 2 - 3 tokens
# (VPTQ_SLICED_ONE_LAUNCH=1: also the layers the size rules keep on the gather kernel - window parts and column phases on small layers)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r5s28; mkdir -p $OUT
cd $R
VPTQ_SLICED_ONE_LAUNCH=1 timeout 300 python tools/gpu_sliced_count.py --layers 1600 --seed 7 2>&1 | grep -v amdgpu.ids | grep -v "layers after" > $OUT/sliced_count_forced_f16.txt; tail -22 $OUT/sliced_count_forced_f16.txt
VPTQ_SLICED_ONE_LAUNCH=1 timeout 300 python tools/gpu_sliced_count.py --layers 800 --seed 8 --dtype bf16 2>&1 | grep -v amdgpu.ids | grep -v "layers after" > $OUT/sliced_count_forced_bf16.txt; tail -3 $OUT/sliced_count_forced_bf16.txt
