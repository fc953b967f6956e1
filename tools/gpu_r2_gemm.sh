#!/bin/bash
OUT=gpurun_out/r2d; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 | tee $OUT/tests.txt
timeout 300 python tools/tokens_bench.py --out $OUT/tokens_new.json 2>&1 | grep "^{" | tee $OUT/tokens_new.txt
VPTQ_GEMM_MIN_TOKENS=99 timeout 300 python tools/tokens_bench.py --shapes "8192,8192" --tokens 5,8,16 2>&1 | grep "^{" | tee $OUT/tokens_old.txt
VPTQ_GEMM_MIN_TOKENS=2 timeout 300 python tools/tokens_bench.py --shapes "8192,8192;4096,4096" --tokens 2,3,4 2>&1 | grep "^{" | tee $OUT/tokens_min2.txt
