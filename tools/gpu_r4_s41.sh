#!/bin/bash
# round 4, step 41: 5 - 8 tokens over the sliced layouts (8 token slots on the matrix pipe): parity, timing at 4096 columns
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4s41; mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gemv_sliced_gpu.py -m gpu -q -p no:cacheprovider --tb=short -x -k "tokens" 2>&1 | tail -12 | tee $OUT/tests.txt
for cfg in "8 256" "8 65536" "16 65536"; do
  set -- $cfg
  timeout 150 python tools/sliced_tokens_bench.py --v $1 --kr $2 --shapes "4096,4096;4096,14336" --only-one-launch 2>&1 | grep -v amdgpu.ids | tee -a $OUT/timing.txt
  timeout 150 python tools/sliced_tokens_bench.py --v $1 --kr $2 --shapes "4096,0" --siblings 4096,1024,1024 2>&1 | grep -v amdgpu.ids | tee -a $OUT/timing.txt
done
