#!/bin/bash
# every vector length + outlier codebooks of vector length 4 on gemv_gatherx: parity, fuzz, timing vs the generic kernel
OUT=gpurun_out/r5a; mkdir -p $OUT
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "gatherx or golden or gemv_and_dequant or token_counts" 2>&1 | tail -15 | tee $OUT/tests.txt
timeout 600 python tools/gpu_fuzz.py --formats --cases 120 --seed 77 2>&1 | tail -8 | tee $OUT/fuzz_f16.txt
timeout 600 python tools/gpu_fuzz.py --formats --cases 60 --seed 78 --dtype bf16 2>&1 | tail -5 | tee $OUT/fuzz_bf16.txt
timeout 600 python tools/format_bench.py --formats v6-k4096-0,v6-k65536-256,v4-k65536-0,v4-k256-256,v10-k65536-1024,v2-k256-0,v8-k65536-256-o128,v8-k65536-0-o256,v16-k65536-65536-o128,v12-k65536-4096-o128 --out $OUT/formats2_8192.json 2>&1 | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('%-24s T=%2d  %-20s %7.1f us %6.0f GB/s | generic %7.1f us %5.0f GB/s | diff %.1e' % (d['format'], d['T'], d['default']['kernel'], d['default']['us_per_launch'], d['default']['GBps'], d['generic']['us_per_launch'], d['generic']['GBps'], d['max_rel_diff_default_vs_generic']))" | tee $OUT/formats2_8192.txt
