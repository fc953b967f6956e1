#!/bin/bash
# fused GEMV vs dequant + dense GEMM per format and token count (where should vptq_quant_gemv_max_tokens sit?)
OUT=gpurun_out/r5i; mkdir -p $OUT; rm -f $OUT/format_tokens.txt
for T in 2 4 5 8; do
  timeout 600 python tools/format_bench.py --dense --tokens $T --formats v8-k65536-256,v8-k65536-0,v16-k65536-65536,v16-k65536-1024,v8-k8192-256,v6-k4096-0 2>&1 | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('tokens %d %-20s %-22s %7.1f us | dequant + F.linear %7.1f us' % ($T, d['format'], d['default']['kernel'], d['default']['us_per_launch'], d['dense']['us_per_launch']))" | tee -a $OUT/format_tokens.txt
done
