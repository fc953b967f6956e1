#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
timeout 300 python tools/microbench.py --hidden 8192 --prefetch --out gpurun_out/mb5_8192_pf.json 2>&1 | grep -E "^(exact|fast|Traceback|Assert)" 
timeout 300 python tools/microbench.py --hidden 8192 --out gpurun_out/mb5_8192.json 2>&1 | grep -E "^(exact|fast|Traceback|Assert)" 
timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c1-300
timeout 300 python bench.py --no-cpu-baseline --no-prefetch 2>&1 | tail -1 | cut -c1-300
timeout 300 python bench.py --no-cpu-baseline --hidden 4096 2>&1 | tail -1 | cut -c1-300
timeout 300 python bench.py --no-cpu-baseline --no-prefetch --hidden 4096 2>&1 | tail -1 | cut -c1-300
