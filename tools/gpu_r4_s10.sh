#!/bin/bash
# round 4, GPU call 10: small residual tables split by column ranges; family tests again
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4s10; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gemv_sliced_gpu.py -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -6 | tee $OUT/gpu_tests.txt
S="8192,8192;4096,4096;28672,8192"
for f in "8 65536 4" "16 65536 256" "16 65536 1024" "16 65536 64"; do
  set -- $f
  timeout 200 python tools/sliced_bench.py --v $1 --k $2 --kr $3 --shapes "$S" 2>&1 | grep -v amdgpu.ids | cut -c1-330 | tee -a $OUT/sliced_family_small_tables.txt
done
