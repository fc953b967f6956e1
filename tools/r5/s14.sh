#!/bin/bash
# round 5, session 14: 2 / 3 tokens in the reference's roundings in ONE pass of the one-token kernel (gemv_sliced.hip, TOK) against
# the column-phase kernel (VPTQ_SLICED_ONE_PASS=0) and the gather kernel
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r5s14; mkdir -p $OUT; rm -f $OUT/*.txt
cd $R
timeout 1200 python -m pytest tests/test_gemv_sliced_gpu.py -x -q -m gpu -k "tokens or wide_layers or rejections" 2>&1 | tail -25 > $OUT/tests.txt; tail -5 $OUT/tests.txt
for op in 1 0; do
for a in "--kr 256" "--kr 0" "--v 16 --kr 0"; do
  echo "== VPTQ_SLICED_ONE_PASS=$op $a" >> $OUT/tok.txt
  VPTQ_SLICED_ONE_PASS=$op timeout 300 python tools/sliced_tokens_exact_bench.py $a --tokens 2,3 --shapes "8192,8192;8192,28672;8192,1024;2048,8192" 2>&1 | grep -v amdgpu.ids >> $OUT/tok.txt
done
done
cat $OUT/tok.txt
