#!/bin/bash
# round 4, step 29: the matrix-pipe contraction for vector length 16 too: parity, timing
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4s29; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gemv_sliced_gpu.py -m gpu -q -p no:cacheprovider --tb=short -k tokens 2>&1 | tail -25 | tee $OUT/tests.txt
for cfg in "16 65536" "16 0" "16 1024"; do
  set -- $cfg
  timeout 200 python tools/sliced_tokens_bench.py --v $1 --kr $2 --shapes "8192,8192;4096,4096;14336,4096;4096,14336" --only-one-launch 2>&1 | grep -v amdgpu.ids | tee -a $OUT/timing.txt
done
