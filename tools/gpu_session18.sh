#!/bin/bash
for d in 0 1 2; do
 for H in 8192 4096; do
  echo "debug=$d H=$H"
  VPTQ_K256_DEBUG=$d timeout 300 python tools/microbench.py --hidden $H 2>&1 | grep -E "^exact" | cut -c1-260
 done
done
