#!/bin/bash
# dequant kernel per format: parity (bit-exact tests) + vptq_dequant + F.linear at 2 tokens (the GEMM part is ~10 us)
OUT=gpurun_out/r5k; mkdir -p $OUT
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "dequant or golden or token_counts or prefill or gemm" 2>&1 | tail -5 | tee $OUT/tests.txt
timeout 600 python tools/format_bench.py --dense --tokens 2 --formats v8-k65536-256,v8-k65536-0,v16-k65536-65536,v16-k65536-1024,v8-k8192-256,v6-k4096-0,v12-k65536-4096,v8-k65536-256-c2 2>&1 | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('%-20s T=%2d dequant + F.linear (2 tokens) %7.1f us' % (d['format'], d['T'], d['dense']['us_per_launch']))" | tee $OUT/dequant_formats.txt
