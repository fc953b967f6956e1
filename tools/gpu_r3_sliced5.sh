#!/bin/bash
# round 3: GEMV over the sliced layout, both formats: parity, then us per layer against the gather kernel
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r3sl; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gemv_sliced_gpu.py -x -q 2>&1 | tail -4 | tee $OUT/tests.txt
timeout 600 python tools/sliced_bench.py --ring 6 --kr 0 --shapes "8192,8192;4096,4096;4096,14336;14336,4096;28672,8192" --out $OUT/sliced_k65536_r0.json 2>&1 | grep -v amdgpu.ids | cut -c1-330 | tee $OUT/sliced_final.txt
timeout 600 python tools/sliced_bench.py --ring 6 --kr 256 --shapes "8192,8192;4096,4096;4096,14336;14336,4096;28672,8192" --out $OUT/sliced_k65536_r256.json 2>&1 | grep -v amdgpu.ids | cut -c1-330 | tee -a $OUT/sliced_final.txt
timeout 600 python tools/sliced_bench.py --ring 6 --kr 256 --shapes "8192,8192" --bf16 --out $OUT/sliced_k65536_r256_bf16.json 2>&1 | grep -v amdgpu.ids | cut -c1-330 | tee -a $OUT/sliced_final.txt
VPTQ_SLICED_LAYOUT=1 timeout 900 python tools/llama_decode.py --k 65536 --kr 256 --new 64 --out $OUT/llama8b_k65536_r256_sliced.json 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-400 | tee $OUT/llama_k65536_sliced.txt
