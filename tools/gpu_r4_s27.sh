#!/bin/bash
# round 4, step 27: the 2 - 4 token sliced kernel with slim scalar bookkeeping per step: parity, timing
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4s27; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gemv_sliced_gpu.py -m gpu -q -p no:cacheprovider --tb=short -x -k tokens 2>&1 | tail -5 | tee $OUT/tests.txt
for cfg in "8 0" "8 256" "8 65536" "16 65536"; do
  set -- $cfg
  timeout 200 python tools/sliced_tokens_bench.py --v $1 --kr $2 --shapes "8192,8192;4096,4096;14336,4096" --only-one-launch 2>&1 | grep -v amdgpu.ids | tee -a $OUT/timing.txt
done
