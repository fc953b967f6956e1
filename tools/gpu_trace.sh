#!/bin/bash
# phase timelines (make -C vptq_amd/csrc trace): HBM-cold ring and L2-hot; H = 8192 by default
OUT=gpurun_out/r3q; mkdir -p $OUT
H=${1:-8192}
for k in mfma valu; do
timeout 300 python tools/trace_k256m.py --hidden $H --kernel $k 2>&1 | grep -v amdgpu.ids | tee $OUT/trace_${H}_${k}_cold.txt
timeout 300 python tools/trace_k256m.py --hidden $H --kernel $k --hot 2>&1 | grep -v amdgpu.ids | tee $OUT/trace_${H}_${k}_hot.txt
done
