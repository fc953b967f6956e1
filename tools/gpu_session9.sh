#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q 2>&1 | grep -E "AssertionError|assert |passed|failed" | head -20
timeout 300 python tools/microbench.py --hidden 8192 --prefetch --out gpurun_out/mb6_8192_pf.json 2>&1 | grep -E "^(exact|fast|Traceback|Assert)" 
timeout 300 python tools/microbench.py --hidden 4096 --prefetch --out gpurun_out/mb6_4096_pf.json 2>&1 | grep -E "^(exact|fast|Traceback|Assert)" 
