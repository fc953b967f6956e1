#!/bin/bash
# round 4, step 35: 2 tokens through the 4-slot matrix-pipe kernel (8 bytes of activations per column: twice the phases) against
# the 2-slot packed-FMA kernel
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4s35; mkdir -p $OUT
cd $R
for t4 in 0 1; do
  echo "== VPTQ_SLICED_TOK4=$t4" | tee -a $OUT/timing.txt
  for cfg in "8 0" "8 256" "8 65536" "16 65536" "16 0"; do
    set -- $cfg
    VPTQ_SLICED_TOK4=$t4 timeout 200 python tools/sliced_tokens_bench.py --v $1 --kr $2 --shapes "8192,8192;4096,4096;14336,4096;4096,14336" --only-one-launch 2>&1 | grep -v amdgpu.ids | tee -a $OUT/timing.txt
  done
done
