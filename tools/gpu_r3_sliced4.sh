#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r3sl; mkdir -p $OUT
cd $R
for v in "" ab1 ab2 ab4 ab3 ab7; do
lib=""; [ -n "$v" ] && lib=$R/tools/_build/libvptq_hip_$v.so
echo "== ${v:-full}" | tee -a $OUT/sliced_ablate.txt
VPTQ_HIP_LIB=$lib timeout 600 python tools/sliced_bench.py --ring 6 --shapes "8192,8192" 2>&1 | grep -v amdgpu.ids | cut -c1-215 | tee -a $OUT/sliced_ablate.txt
done
