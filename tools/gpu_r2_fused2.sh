#!/bin/bash
OUT=gpurun_out/r2h; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q -k "fused" 2>&1 | tail -3
timeout 600 python tools/prefill_bench.py --tokens ${TOKENS:-256,1024,8192} --shapes "8192,8192" --dtypes ${DTYPES:-f16} 2>&1 | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('%s M=%5d fused %8.1f us %6.0f TF (%.3f)  dense %8.1f us  ratio %.2f  diff %.1e' % (d['dtype'], d['tokens'], d['fused_us'], d['fused_TFLOPs'], d['fused_frac_of_2500TF'], d['dense_us'], d['fused_vs_dense'], d['rel_diff']))"
