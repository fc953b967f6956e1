#!/bin/bash
# round 4, GPU call 9: the rest of the k >= 16384 family over sliced layouts (any residual size as a second table): parity + timing
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4s9; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gemv_sliced_gpu.py tests/test_hip_parity.py -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -15 | tee $OUT/gpu_tests.txt
S="8192,8192;4096,4096;28672,8192"
for f in "16 65536 1024" "16 65536 16384" "16 65536 256" "8 65536 4" "8 65536 4096" "8 32768 0" "8 65536 65536" "8 65536 256"; do
  set -- $f
  timeout 200 python tools/sliced_bench.py --v $1 --k $2 --kr $3 --shapes "$S" 2>&1 | grep -v amdgpu.ids | cut -c1-330 | tee -a $OUT/sliced_family.txt
done
