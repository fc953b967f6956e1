#!/bin/bash
# round 5, session 13: 2 - 8 tokens in the reference's roundings over the exact sliced layouts (gemv_sliced_tok.hip, EX)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r5s13; mkdir -p $OUT; rm -f $OUT/*.txt
cd $R
timeout 1200 python -m pytest tests/test_gemv_sliced_gpu.py -x -q -m gpu 2>&1 | tail -5 > $OUT/tests.txt; cat $OUT/tests.txt
for a in "--kr 256" "--kr 0"; do
  echo "== $a" >> $OUT/tok.txt
  timeout 300 python tools/sliced_tokens_exact_bench.py $a --shapes "8192,8192;14336,4096;8192,28672" 2>&1 | grep -v amdgpu.ids >> $OUT/tok.txt
done
cat $OUT/tok.txt
