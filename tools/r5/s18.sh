#!/bin/bash
# round 5, session 18: layers too wide for the reference's roundings in one piece as COLUMN PARTS (28672 columns = 2 x 14336): tests,
# fuzz, timings against the gather kernel
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r5s18; mkdir -p $OUT; rm -f $OUT/*.txt
cd $R
timeout 1200 python -m pytest tests/test_gemv_sliced_gpu.py -x -q -m gpu -k "reference_roundings" 2>&1 | tail -15 > $OUT/tests.txt; tail -4 $OUT/tests.txt
timeout 600 python tools/gpu_fuzz.py --sliced --cases 40 --seed 77 2>&1 | grep -v amdgpu.ids > $OUT/fuzz_sliced_f16.txt; tail -3 $OUT/fuzz_sliced_f16.txt | cut -c1-300
for a in "--kr 256" "--kr 0" "--kr 65536" "--v 16 --kr 0"; do
  echo "== $a" >> $OUT/parts.txt
  timeout 300 python tools/sliced_bench.py --exact $a --shapes "28672,8192;28672,1024;24576,6144" --ring 4 2>&1 | grep -v amdgpu.ids | cut -c1-420 >> $OUT/parts.txt
done
cat $OUT/parts.txt
