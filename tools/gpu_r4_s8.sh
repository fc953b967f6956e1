#!/bin/bash
# round 4, GPU call 8: both tables of the 65536-residual formats in ONE launch: parity + timing; 4-bit decode again
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4s8; mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gemv_sliced_gpu.py tests/test_hip_parity.py -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -8 | tee $OUT/gpu_tests.txt
S="8192,8192;4096,4096;4096,1024;4096,14336;14336,4096;28672,8192"
timeout 200 python tools/sliced_bench.py --kr 65536 --shapes "$S" --out $OUT/sliced_k65536_r65536.json 2>&1 | grep -v amdgpu.ids | tee $OUT/sliced_k65536_r65536.txt
timeout 200 python tools/sliced_bench.py --v 16 --kr 65536 --shapes "$S" --out $OUT/sliced_v16_k65536_r65536.json 2>&1 | grep -v amdgpu.ids | tee $OUT/sliced_v16_k65536_r65536.txt
VPTQ_SLICED_LAYOUT=1 timeout 300 python tools/llama_decode.py --fuse --k 65536 --kr 65536 --new 128 --out $OUT/llama8b_k65536_r65536_sliced.json 2> /dev/null | tail -1 | cut -c1-700 | tee $OUT/llama8b_k65536_r65536_sliced.txt
