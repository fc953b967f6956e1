#!/bin/bash
OUT=gpurun_out/r2b; mkdir -p $OUT
for v in afnt afntx ntx; do
  echo "== parity with $v"; VPTQ_HIP_LIB=$PWD/tools/_build/libvptq_hip_$v.so timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $OUT/tests_$v.txt
done
for rep in 1 2; do
  for v in base addform afnt afntx ntx; do
    lib=$PWD/tools/_build/libvptq_hip_$v.so; [ $v = base ] && lib=$PWD/vptq_amd/libvptq_hip.so
    echo "-- $v rep $rep"
    VPTQ_HIP_LIB=$lib timeout 200 python tools/microbench.py --hidden 8192 --group 4 --variants default --no-copy 2>&1 | grep "^default" | tee -a $OUT/ab_$v.txt
  done
done
