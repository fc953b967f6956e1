#!/bin/bash
OUT=gpurun_out/r5l; mkdir -p $OUT; rm -f $OUT/wide.txt
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "gatherx" 2>&1 | tail -4 | tee $OUT/tests.txt
for rep in 1 2; do for w in 1 0; do
  VPTQ_GATHERX_WIDE=$w timeout 300 python tools/format_bench.py --formats v16-k65536-65536,v16-k65536-1024,v16-k65536-0,v12-k65536-4096 2>&1 | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('wide=$w %-20s %-20s %7.1f us %6.0f GB/s' % (d['format'], d['default']['kernel'], d['default']['us_per_launch'], d['default']['GBps']))" | tee -a $OUT/wide.txt
done; done
