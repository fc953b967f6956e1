#!/bin/bash
OUT=gpurun_out/r2h; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q -k "fused" 2>&1 | tail -25 | tee $OUT/tests_fused.txt
timeout 600 python tools/prefill_bench.py --tokens 64,128,512,1024,8192 --shapes "8192,8192" --dtypes f16,bf16 --out $OUT/prefill.json 2>&1 | grep "^{" | tee $OUT/prefill.txt
