#!/bin/bash
# round 4, GPU call 14: 2-4 tokens: gather kernels against one sliced launch per token
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4s14; mkdir -p $OUT
cd $R
S="8192,8192;4096,4096;4096,14336"
for f in "8 0" "8 256" "8 65536" "16 65536" "16 1024"; do
  set -- $f
  timeout 300 python tools/sliced_tokens_bench.py --v $1 --kr $2 --shapes "$S" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/sliced_tokens.txt
done
