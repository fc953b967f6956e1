#!/bin/bash
# round 4, step 19: 2 - 4 tokens over the sliced layouts in one launch (gemv_sliced_tok.hip) - parity, then timing; the one-token
# kernel over the window-ordered layouts (before: profiles/r04/sliced_family.txt, sliced_tokens.txt)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4s19; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gemv_sliced_gpu.py -m gpu -q -p no:cacheprovider --tb=short -x 2>&1 | tail -40 | tee $OUT/tests.txt
for cfg in "8 0" "8 256" "8 65536" "16 65536"; do
  set -- $cfg
  timeout 300 python tools/sliced_tokens_bench.py --v $1 --kr $2 --shapes "8192,8192;4096,4096;4096,14336;14336,4096" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/sliced_tokens.txt
done
