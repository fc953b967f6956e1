#!/bin/bash
# round 4, step 25: 2 - 4 tokens over the sliced layouts in one launch, routed from VQuantLinear.forward: the GPU suite, the
# family fuzzer with random token counts, the timing table
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4s25; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --tb=short 2>&1 | tail -30 | tee $OUT/suite.txt
timeout 400 python tools/gpu_fuzz.py --sliced --cases 60 --seed 1601 2>&1 | tail -4 | tee $OUT/fuzz_sliced.txt
timeout 300 python tools/gpu_fuzz.py --sliced --cases 30 --seed 1602 --dtype bf16 2>&1 | tail -2 | tee -a $OUT/fuzz_sliced.txt
for cfg in "8 0" "8 256" "8 65536" "16 65536" "16 0"; do
  set -- $cfg
  timeout 300 python tools/sliced_tokens_bench.py --v $1 --kr $2 --shapes "8192,8192;4096,4096;4096,14336;14336,4096" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/sliced_tokens_one_launch.txt
done
