#!/bin/bash
OUT=gpurun_out/r2c; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 | tee $OUT/tests.txt
for k in "8192 256" "8192 512" "4096 256" "4096 0"; do set -- $k
  timeout 300 python tools/microbench.py --hidden 8192 --k $1 --kr $2 --no-copy 2>&1 | grep -E "^(default|generic) " | sed "s/^/k=$1 kr=$2 /" | tee -a $OUT/mb_lds.txt
done
timeout 300 python tools/microbench.py --hidden 4096 --k 8192 --kr 256 --no-copy 2>&1 | grep -E "^(default|generic) " | sed "s/^/H=4096 k=8192 kr=256 /" | tee -a $OUT/mb_lds.txt
