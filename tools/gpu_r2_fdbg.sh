#!/bin/bash
for v in base fdbg1 fdbg2 fdbg3; do
  lib=$PWD/tools/_build/libvptq_hip_$v.so; [ $v = base ] && lib=$PWD/vptq_amd/libvptq_hip.so
  echo "== $v"; VPTQ_HIP_LIB=$lib TOKENS=8192 DTYPES=f16 bash tools/gpu_r2_fused2.sh 2>&1 | grep "M="
done
