#!/bin/bash
# round 4, step 30: the full table for 2 - 8 tokens: gather kernels / one sliced launch per token / one launch (2 - 4 tokens) /
# two launches (6, 8 tokens)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4s30; mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gemv_sliced_gpu.py -m gpu -q -p no:cacheprovider --tb=short -k tokens 2>&1 | tail -5 | tee $OUT/tests.txt
for cfg in "8 0" "8 256" "8 65536" "16 65536" "16 0" "16 1024" "8 4096"; do
  set -- $cfg
  timeout 300 python tools/sliced_tokens_bench.py --v $1 --kr $2 --shapes "8192,8192;4096,4096;4096,14336;14336,4096" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/sliced_tokens_one_launch.txt
done
