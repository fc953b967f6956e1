#!/bin/bash
# round 5, session 24: counting fuzz of the large-codebook formats through VQuantLinear.forward (default arithmetic) on checkpoint-like
# families and spiky activations, 1 - 3 tokens: the round's sliced routes (one token, column parts, RG, one pass for 2 / 3 tokens)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r5s24; mkdir -p $OUT
cd $R
timeout 420 python tools/gpu_sliced_count.py --layers 2400 --seed 3 2>&1 | grep -v amdgpu.ids | grep -v "layers after" > $OUT/sliced_count_f16.txt; tail -24 $OUT/sliced_count_f16.txt
timeout 300 python tools/gpu_sliced_count.py --layers 1200 --seed 4 --dtype bf16 2>&1 | grep -v amdgpu.ids | grep -v "layers after" > $OUT/sliced_count_bf16.txt; tail -22 $OUT/sliced_count_bf16.txt
