#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q 2>&1 | grep -E "AssertionError|assert |passed|failed|Error" | head -20
for kr in 0 256 65536; do
  echo "k=65536 kr=$kr"
  timeout 300 python tools/microbench.py --hidden 8192 --k 65536 --kr $kr --ring 16 --out gpurun_out/mb_alt_8192_kr$kr.json 2>&1 | grep -E "^(exact|generic|Traceback|Assert)" | cut -c1-300
done
