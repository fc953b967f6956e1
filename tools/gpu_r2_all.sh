#!/bin/bash
OUT=gpurun_out/r2i; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee $OUT/tests.txt
timeout 900 python tools/prefill_bench.py --tokens 64,128,256,512,1024,2048,8192 --shapes "4096,4096;8192,8192" --dtypes f16,bf16 --out $OUT/prefill_fused_vs_dense.json 2>&1 | grep -c "^{"
