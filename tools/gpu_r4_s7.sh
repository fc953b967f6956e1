#!/bin/bash
# round 4, GPU call 7: full GPU suite; Llama-3-8B-shaped decode in the 4-bit format v8-k65536-65536 with / without the sliced layouts
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4s7; mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -12 | tee $OUT/gpu_suite.txt
timeout 300 python tools/llama_decode.py --fuse --k 65536 --kr 65536 --new 128 --out $OUT/llama8b_k65536_r65536_sliced.json 2> /dev/null | tail -1 | cut -c1-700 | tee $OUT/llama8b_k65536_r65536_sliced.txt
VPTQ_SLICED_LAYOUT=0 timeout 300 python tools/llama_decode.py --fuse --k 65536 --kr 65536 --new 128 --out $OUT/llama8b_k65536_r65536_default.json 2> /dev/null | tail -1 | cut -c1-700 | tee $OUT/llama8b_k65536_r65536_default.txt
