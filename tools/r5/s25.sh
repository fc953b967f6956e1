#!/bin/bash
# round 5, session 25: 2 / 3 tokens in one pass in WINDOW PARTS (layers whose slice leaves room for half of the columns' operands: 4096
# columns beside 128 KiB, 14336 beside 64 KiB): tests, timings against the column-phase kernel (VPTQ_SLICED_WINDOW_PARTS=0) and the gather kernel
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r5s25; mkdir -p $OUT; rm -f $OUT/*.txt
cd $R
timeout 900 python -m pytest tests/test_gemv_sliced_gpu.py -x -q -m gpu -k "tokens_reference_roundings or rejections" 2>&1 | tail -12 > $OUT/tests.txt; tail -5 $OUT/tests.txt
for wp in 1 0; do
for a in "--kr 256" "--kr 0"; do
  echo "== VPTQ_SLICED_WINDOW_PARTS=$wp $a" >> $OUT/tok.txt
  VPTQ_SLICED_WINDOW_PARTS=$wp timeout 300 python tools/sliced_tokens_exact_bench.py $a --tokens 2,3 --shapes "4096,4096;4096,14336;14336,4096;4096,1024" 2>&1 | grep -v amdgpu.ids >> $OUT/tok.txt
done
done
cat $OUT/tok.txt
