#!/bin/bash
# round 4, GPU call 6: sliced layout for the two-table format and vector length 16: parity tests, timing against the gather kernels
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4s6; mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gemv_sliced_gpu.py tests/test_hip_parity.py -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -25 | tee $OUT/gpu_tests.txt
S="8192,8192;4096,4096;4096,14336;14336,4096;28672,8192"
timeout 200 python tools/sliced_bench.py --kr 65536 --shapes "$S" --out $OUT/sliced_k65536_r65536.json 2>&1 | grep -v amdgpu.ids | tee $OUT/sliced_k65536_r65536.txt
timeout 200 python tools/sliced_bench.py --v 16 --kr 65536 --shapes "$S" --out $OUT/sliced_v16_k65536_r65536.json 2>&1 | grep -v amdgpu.ids | tee $OUT/sliced_v16_k65536_r65536.txt
timeout 200 python tools/sliced_bench.py --v 16 --kr 0 --shapes "$S" --out $OUT/sliced_v16_k65536_r0.json 2>&1 | grep -v amdgpu.ids | tee $OUT/sliced_v16_k65536_r0.txt
