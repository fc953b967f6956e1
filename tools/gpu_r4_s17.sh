#!/bin/bash
# round 4, GPU call 17: sibling layers in one sliced launch: tests + Llama-3-8B-shaped decode in the k65536 formats
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4s17; mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gemv_sliced_gpu.py -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -12 | tee $OUT/gpu_tests.txt
for kr in 256 0 65536; do
  timeout 300 python tools/llama_decode.py --fuse --k 65536 --kr $kr --new 128 2>/dev/null | tail -1 | cut -c1-800 | tee $OUT/llama8b_k65536_r${kr}_grouped.json
done
