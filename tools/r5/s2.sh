#!/bin/bash
# round 5, session 2: raw phase stamps of gemv_sliced (who are the stragglers?)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r5s2; mkdir -p $OUT
cd $R
for kr in 0 256; do
  VPTQ_HIP_LIB=$R/tools/_build/libvptq_hip_tr1.so timeout 120 python tools/sliced_trace.py --kr $kr --raw $OUT/trace_kr$kr.npy 2>&1 | grep -v amdgpu.ids > $OUT/trace_kr$kr.json
done
VPTQ_HIP_LIB=$R/tools/_build/libvptq_hip_tr1.so timeout 120 python tools/sliced_trace.py --kr 0 --shape 4096,4096 --raw $OUT/trace_kr0_4096.npy 2>&1 | grep -v amdgpu.ids > $OUT/trace_kr0_4096.json
ls -la $OUT
