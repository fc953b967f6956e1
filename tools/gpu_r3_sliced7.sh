#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r3sl7; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gemv_sliced_gpu.py -x -q 2>&1 | tail -4 | tee $OUT/tests.txt
for i in 1 2 3; do timeout 300 python -m pytest tests/test_gemv_sliced_gpu.py -x -q -k "vs_oracle or goldens" 2>&1 | tail -1; done | tee -a $OUT/tests.txt
timeout 600 python tools/sliced_bench.py --ring 6 --kr 0 --shapes "8192,8192;4096,4096;4096,14336;14336,4096;28672,8192" --out $OUT/sliced_k65536_r0.json 2>&1 | grep -v amdgpu.ids | cut -c1-250 | tee $OUT/sliced_final.txt
timeout 600 python tools/sliced_bench.py --ring 6 --kr 256 --shapes "8192,8192;4096,4096;4096,14336;14336,4096;28672,8192" --out $OUT/sliced_k65536_r256.json 2>&1 | grep -v amdgpu.ids | cut -c1-250 | tee -a $OUT/sliced_final.txt
timeout 600 python tools/sliced_bench.py --ring 6 --kr 256 --shapes "8192,8192" --bf16 --out $OUT/sliced_k65536_r256_bf16.json 2>&1 | grep -v amdgpu.ids | cut -c1-250 | tee -a $OUT/sliced_final.txt
timeout 900 python tools/llama_decode.py --k 65536 --kr 256 --new 64 --out $OUT/llama8b_k65536_r256_sliced.json 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-400 | tee $OUT/llama_k65536_sliced.txt
