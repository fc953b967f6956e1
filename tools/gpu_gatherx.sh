#!/bin/bash
OUT=gpurun_out/r3n; mkdir -p $OUT
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "gatherx or golden or gemv_and_dequant or token_counts" 2>&1 | tail -15 | tee $OUT/tests.txt
timeout 600 python tools/format_bench.py --out $OUT/formats_8192.json 2>&1 | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('%-22s T=%2d  %-20s %7.1f us %6.0f GB/s | generic %7.1f us %5.0f GB/s | diff %.1e' % (d['format'], d['T'], d['default']['kernel'], d['default']['us_per_launch'], d['default']['GBps'], d['generic']['us_per_launch'], d['generic']['GBps'], d['max_rel_diff_default_vs_generic']))" | tee $OUT/formats_8192.txt
