#!/bin/bash
# phase timeline of gemv_k256m_kernel (make -C vptq_amd/csrc trace), HBM-cold ring and L2-hot
OUT=gpurun_out/r3j; mkdir -p $OUT
timeout 300 python tools/trace_k256m.py --hidden 8192 2>&1 | grep -v amdgpu.ids | tee $OUT/trace_8192_cold.txt
timeout 300 python tools/trace_k256m.py --hidden 8192 --hot 2>&1 | grep -v amdgpu.ids | tee $OUT/trace_8192_hot.txt
