#!/bin/bash
OUT=gpurun_out/r2g; mkdir -p $OUT
( time timeout 900 python bench.py ) 2>&1 | tail -6 | tee $OUT/bench_default.txt
timeout 600 python bench.py --mode tp_row --tp-layers 20 --no-extras 2>&1 | tail -1 | tee $OUT/bench_tp_row_n1.json
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee $OUT/tests.txt
