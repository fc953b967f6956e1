#!/bin/bash
# round 5, session 15: VPTQ_SLICED_SLICES=16 (the exact layouts of 4096-column layers with 16 instead of 8 slices): what the one-pass
# kernel for 2 / 3 tokens gives there, against what it costs one token
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r5s15; mkdir -p $OUT; rm -f $OUT/*.txt
cd $R
for s in 8 16; do
  echo "== VPTQ_SLICED_SLICES=$s" >> $OUT/tok.txt
  for a in "--kr 256" "--kr 0"; do
    VPTQ_SLICED_SLICES=$s timeout 300 python tools/sliced_tokens_exact_bench.py $a --tokens 2,3,4 --shapes "4096,4096;4096,14336;4096,6144" 2>&1 | grep -v amdgpu.ids >> $OUT/tok.txt
    VPTQ_SLICED_SLICES=$s timeout 300 python tools/sliced_bench.py --exact $a --shapes "4096,4096;4096,14336;4096,6144" 2>&1 | grep -v amdgpu.ids | cut -c1-330 >> $OUT/tok.txt
  done
done
cat $OUT/tok.txt
