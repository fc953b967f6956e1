#!/bin/bash
# round 4, step 32: matrix-pipe mode with the rows' sums in registers (<= 4 rows per wave): parity, timing against LDS sums
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4s32; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gemv_sliced_gpu.py -m gpu -q -p no:cacheprovider --tb=short -k tokens 2>&1 | tail -8 | tee $OUT/tests.txt
for lds in 0 1; do
  echo "== VPTQ_SLICED_LDS_SUMS=$lds" | tee -a $OUT/timing.txt
  for cfg in "8 0" "8 256" "8 65536" "16 65536" "16 0"; do
    set -- $cfg
    VPTQ_SLICED_LDS_SUMS=$lds timeout 200 python tools/sliced_tokens_bench.py --v $1 --kr $2 --shapes "8192,8192;4096,4096;14336,4096;4096,14336" --only-one-launch 2>&1 | grep -v amdgpu.ids | tee -a $OUT/timing.txt
  done
done
