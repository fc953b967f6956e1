#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r3n; mkdir -p $OUT
cd $R
B=$R/tools/_build
for v in prof prioprof profabl; do
  echo "== $v" | tee -a $OUT/chain_prof2.txt
  VPTQ_HIP_LIB=$B/libvptq_hip_$v.so VPTQ_K256C_PROF=1 timeout 300 python tools/chain_prof.py 2>&1 | grep -v amdgpu.ids | tail -16 | tee -a $OUT/chain_prof2.txt
done
