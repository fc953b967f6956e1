#!/bin/bash
# round 4, step 36: sibling projections in one launch of the token kernel: parity, q / k / v and gate / up of an 8B model
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4s36; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gemv_sliced_gpu.py -m gpu -q -p no:cacheprovider --tb=short -k "tokens or sibling" 2>&1 | tail -12 | tee $OUT/tests.txt
for cfg in "8 256" "8 65536" "16 65536"; do
  set -- $cfg
  timeout 200 python tools/sliced_tokens_bench.py --v $1 --kr $2 --shapes "4096,0" --siblings 4096,1024,1024 2>&1 | grep -v amdgpu.ids | tee -a $OUT/siblings.txt
  timeout 200 python tools/sliced_tokens_bench.py --v $1 --kr $2 --shapes "4096,0" --siblings 14336,14336 2>&1 | grep -v amdgpu.ids | tee -a $OUT/siblings.txt
done
