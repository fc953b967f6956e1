#!/bin/bash
OUT=gpurun_out/r2f; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee $OUT/tests.txt
timeout 600 python tools/tokens_crossover.py --shapes "4096,4096;8192,8192;8192,28672" --tokens 8,16,32,48,64,96 --out $OUT/tokens_crossover.json 2>&1 | grep "^{" | tee $OUT/crossover.txt
