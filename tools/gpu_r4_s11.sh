#!/bin/bash
# round 4, GPU call 11: 8 against 16 slices for v = 8 layers (one token), same box
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4s11; mkdir -p $OUT
cd $R
S="8192,8192;4096,4096;4096,14336;14336,4096"
for kr in 0 256 65536; do
  timeout 200 python tools/sliced_bench.py --kr $kr --shapes "$S" 2>&1 | grep -v amdgpu.ids | cut -c1-330 | sed "s/^/slices8  /" | tee -a $OUT/sliced_8_vs_16_slices.txt
  VPTQ_SLICED_SLICES=16 timeout 200 python tools/sliced_bench.py --kr $kr --shapes "$S" 2>&1 | grep -v amdgpu.ids | cut -c1-330 | sed "s/^/slices16 /" | tee -a $OUT/sliced_8_vs_16_slices.txt
done
