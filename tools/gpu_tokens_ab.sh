#!/bin/bash
# gpurun -- 'bash tools/gpu_tokens_ab.sh' : 1 / 2 / 4 tokens, library default vs VALU kernel forced
mkdir -p gpurun_out/tok
S="${SHAPES:-8192,8192;4096,4096;8192,1024;4096,14336;8192,28672}"
for F in 0 16; do
  echo "== flags=$F"
  timeout 400 python tools/shape_bench.py --shapes "$S" --tokens 1,2,3,4 --flags $F --out gpurun_out/tok/shapes_flags$F.json 2>&1 | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('  %-24s %5dx%-5d tok=%d %7.2f us %6.0f GB/s' % (d['kernel'], d['I'], d['O'], d['tokens'], d['us_per_launch'], d['GBps']))"
done
