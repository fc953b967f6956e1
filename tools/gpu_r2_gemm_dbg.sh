#!/bin/bash
OUT=gpurun_out/r2e; mkdir -p $OUT
for d in 0 1 2 3 4 8 12 16 31; do
  echo "dbg=$d"; VPTQ_GEMM_DBG=$d timeout 300 python tools/tokens_bench.py --shapes "8192,8192" --tokens 16 2>&1 | grep "^{" | tee -a $OUT/dbg.txt
done
