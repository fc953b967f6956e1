#!/bin/bash
for rep in 1 2; do for v in base lgb8; do
  lib=$PWD/tools/_build/libvptq_hip_$v.so; [ $v = base ] && lib=$PWD/vptq_amd/libvptq_hip.so
  for k in "8192 256" "4096 256"; do set -- $k
  VPTQ_HIP_LIB=$lib timeout 300 python tools/microbench.py --hidden 8192 --k $1 --kr $2 --variants default --no-copy 2>&1 | grep -E "^default " | sed "s/^/$v k=$1 kr=$2 /"
  done
done; done
