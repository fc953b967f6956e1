#!/bin/bash
for pf in none 4096 8192 16384 65536; do
  echo "pf=$pf"
  if [ $pf = none ]; then
    timeout 300 python tools/microbench.py --hidden 8192 2>&1 | grep -E "^exact" | cut -c1-120
  else
    VPTQ_PF_BYTES=$pf timeout 300 python tools/microbench.py --hidden 8192 --prefetch 2>&1 | grep -E "^exact" | cut -c1-120
  fi
done
