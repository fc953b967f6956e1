#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
for H in 8192 4096; do
  timeout 300 python tools/microbench.py --hidden $H --prefetch --out gpurun_out/mb4_${H}_pf.json 2>&1 | grep -E "^(exact|fast|Traceback|Assert)" 
done
timeout 300 python tools/microbench.py --hidden 8192 --perm --out gpurun_out/mb4_8192_perm.json 2>&1 | grep -E "^(exact|fast)"
timeout 300 python tools/microbench.py --hidden 8192 --perm --prefetch --out gpurun_out/mb4_8192_perm_pf.json 2>&1 | grep -E "^(exact|fast)"
timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c1-400
timeout 300 python bench.py --no-cpu-baseline --no-prefetch 2>&1 | tail -1 | cut -c1-400
