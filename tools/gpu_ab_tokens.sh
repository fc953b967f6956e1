#!/bin/bash
# 2-4 token launches of the MFMA kernel: spill-free build vs the previous one (tools/_build/libvptq_hip_prev.so)
OUT=gpurun_out/r3w; mkdir -p $OUT
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "several_tokens or token_counts or canonical or sibling" 2>&1 | tail -3 | tee $OUT/tests.txt
for rep in 1 2; do
for v in prev new; do
  lib=$PWD/tools/_build/libvptq_hip_prev.so; [ $v = new ] && lib=$PWD/vptq_amd/libvptq_hip.so
  VPTQ_HIP_LIB=$lib timeout 300 python tools/shape_bench.py --shapes "8192,8192;8192,28672;4096,14336;14336,4096" --tokens 2,3,4 2>&1 | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$v rep$rep %5dx%-5d t%d %7.2f us %6.0f GB/s' % (d['I'], d['O'], d['tokens'], d['us_per_launch'], d['GBps']))" | tee -a $OUT/tokens.txt
done; done
