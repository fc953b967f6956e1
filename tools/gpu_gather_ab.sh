#!/bin/bash
OUT=gpurun_out/r3z3; mkdir -p $OUT
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "golden or gemv_and_dequant or token_counts" 2>&1 | tail -3 | tee $OUT/tests.txt
for rep in 1 2; do for v in prev e4 new; do
  lib=$PWD/tools/_build/libvptq_hip_$v.so; [ $v = new ] && lib=$PWD/vptq_amd/libvptq_hip.so
  VPTQ_HIP_LIB=$lib timeout 300 python tools/format_bench.py --formats v8-k65536-256 2>&1 | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$v rep$rep %-18s %-20s %7.1f us %6.0f GB/s' % (d['format'], d['default']['kernel'], d['default']['us_per_launch'], d['default']['GBps']))" | tee -a $OUT/gather.txt
  VPTQ_HIP_LIB=$lib timeout 300 python tools/format_bench.py --hidden 4096 --formats v8-k65536-256 --tokens 2 2>&1 | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$v rep$rep h4096 t2 %-18s %7.1f us' % (d['format'], d['default']['us_per_launch']))" | tee -a $OUT/gather.txt
done; done
