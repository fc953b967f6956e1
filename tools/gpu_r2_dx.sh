#!/bin/bash
OUT=gpurun_out/r2m; mkdir -p $OUT
for v in dx2; do
  echo "== parity $v"; VPTQ_HIP_LIB=$PWD/tools/_build/libvptq_hip_$v.so timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
done
for rep in 1 2 3; do
  for v in base dx2; do
    lib=$PWD/tools/_build/libvptq_hip_$v.so; [ $v = base ] && lib=$PWD/vptq_amd/libvptq_hip.so
    VPTQ_HIP_LIB=$lib timeout 200 python tools/microbench.py --hidden 8192 --group 4 --variants default --no-copy 2>&1 | grep -E "^default " | sed "s/^/$v rep$rep /" | tee -a $OUT/ab_dx.txt
  done
done
for v in base dx2; do
  lib=$PWD/tools/_build/libvptq_hip_$v.so; [ $v = base ] && lib=$PWD/vptq_amd/libvptq_hip.so
  VPTQ_HIP_LIB=$lib timeout 300 python tools/shape_bench.py --model 70b --tokens 1 2>&1 | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$v  %s %5dx%-5d %7.2f us %6.0f GB/s' % (d['kernel'], d['I'], d['O'], d['us_per_launch'], d['GBps']))"
done
