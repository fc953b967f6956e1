#!/bin/bash
# round 4, step 37: sibling groups of the token kernel with at most 4 rows per wave for 3 - 4 tokens; Llama-3-8B-shaped decode is one token (unchanged)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4s37; mkdir -p $OUT
cd $R
for cfg in "8 256" "8 65536" "16 65536" "8 0"; do
  set -- $cfg
  timeout 200 python tools/sliced_tokens_bench.py --v $1 --kr $2 --shapes "4096,0" --siblings 4096,1024,1024 2>&1 | grep -v amdgpu.ids | tee -a $OUT/siblings.txt
  timeout 200 python tools/sliced_tokens_bench.py --v $1 --kr $2 --shapes "4096,0" --siblings 14336,14336 2>&1 | grep -v amdgpu.ids | tee -a $OUT/siblings.txt
done
timeout 600 python -m pytest tests/test_gemv_sliced_gpu.py -m gpu -q -p no:cacheprovider --tb=short -k "sibling" 2>&1 | tail -3 | tee $OUT/tests.txt
