#!/bin/bash
# round 5, session 22: 2 / 3 tokens of the two-table formats of v = 8 (v8-k65536-65536: "4 bit") in the reference's roundings in ONE pass
# of the one-token kernel (RG + TOK: the residual entries are gathered from L2 once for all tokens): tests, timings against the gather kernel
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r5s22; mkdir -p $OUT; rm -f $OUT/*.txt
cd $R
timeout 1200 python -m pytest tests/test_gemv_sliced_gpu.py -x -q -m gpu -k "reference_roundings or rejections or wide_layers" 2>&1 | tail -15 > $OUT/tests.txt; tail -4 $OUT/tests.txt
for a in "--kr 65536" "--kr 4096"; do
  echo "== $a" >> $OUT/tok.txt
  timeout 300 python tools/sliced_tokens_exact_bench.py $a --tokens 2,3 --shapes "8192,8192;8192,28672;8192,1024;2048,8192" 2>&1 | grep -v amdgpu.ids >> $OUT/tok.txt
done
cat $OUT/tok.txt
