#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
for H in 8192 4096; do
  timeout 300 python tools/microbench.py --hidden $H --out gpurun_out/mb3_${H}.json 2>&1 | grep -E "^(exact|fast|Traceback|Assert)" 
done
timeout 300 python tools/microbench.py --hidden 8192 --group 4 --out gpurun_out/mb3_8192_g4.json 2>&1 | grep -E "^(exact|fast)"
timeout 300 python tools/microbench.py --hidden 8192 --perm --out gpurun_out/mb3_8192_perm.json 2>&1 | grep -E "^(exact|fast)"
timeout 300 tools/_build/ubench > gpurun_out/ubench2.txt 2>&1; grep -E "pk_fma_f32|pk_add_f16|fma_mix|v_and|perm" gpurun_out/ubench2.txt
