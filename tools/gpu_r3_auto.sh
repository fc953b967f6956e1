#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r3auto; mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -25 | tee $OUT/gpu_suite.txt
timeout 900 python tools/llama_decode.py --k 65536 --kr 256 --new 64 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-400 | tee $OUT/llama_k65536_auto.txt
timeout 900 python tools/llama_decode.py --fuse --new 64 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-400 | tee $OUT/llama_k256_auto.txt
