#!/bin/bash
OUT=gpurun_out/r3z4; mkdir -p $OUT
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "gatherx or golden or gemv_and_dequant" 2>&1 | tail -3 | tee $OUT/tests.txt
timeout 300 python tools/gpu_fuzz.py --formats --cases 120 --seed 7 2>&1 | tail -1 | tee -a $OUT/tests.txt
for rep in 1 2; do for v in prev new; do
  lib=$PWD/tools/_build/libvptq_hip_$v.so; [ $v = new ] && lib=$PWD/vptq_amd/libvptq_hip.so
  VPTQ_HIP_LIB=$lib timeout 300 python tools/format_bench.py --formats v16-k65536-1024,v8-k65536-1024,v16-k65536-256,v8-k32768-512,v12-k65536-1024 2>&1 | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$v rep$rep %-18s %-20s %7.1f us %6.0f GB/s | generic %6.1f' % (d['format'], d['default']['kernel'], d['default']['us_per_launch'], d['default']['GBps'], d['generic']['us_per_launch']))" | tee -a $OUT/gather.txt
done; done
