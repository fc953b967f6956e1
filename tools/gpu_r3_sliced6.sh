#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r3sl; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gemv_sliced_gpu.py -x -q 2>&1 | tail -6 | tee $OUT/tests.txt
timeout 900 python tools/llama_decode.py --k 65536 --kr 256 --new 64 --out $OUT/llama8b_k65536_r256_default.json 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-400 | tee $OUT/llama_k65536.txt
VPTQ_SLICED_LAYOUT=1 timeout 900 python tools/llama_decode.py --k 65536 --kr 256 --new 64 --out $OUT/llama8b_k65536_r256_sliced.json 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-400 | tee -a $OUT/llama_k65536.txt
