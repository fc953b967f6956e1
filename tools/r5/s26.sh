#!/bin/bash
# round 5, session 26: window parts through the module (tests), and the Llama-3-8B-shaped decode loop in v8-k65536-256 at 1 - 3 sequences with
# them (default) and without (VPTQ_SLICED_WINDOW_PARTS=0)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r5s26; mkdir -p $OUT; rm -f $OUT/*.txt $OUT/*.json
cd $R
timeout 900 python -m pytest tests/test_gemv_sliced_gpu.py -x -q -m gpu -k "wide_layers or sibling or module" 2>&1 | tail -12 > $OUT/tests.txt; tail -4 $OUT/tests.txt
for wp in 1 0; do
  for b in 1 2 3; do
    VPTQ_SLICED_WINDOW_PARTS=$wp timeout 400 python tools/llama_decode.py --fuse --k 65536 --kr 256 --new 128 --batch $b --out $OUT/llama8b_k65536_r256_window_parts_${wp}_batch$b.json > $OUT/llama_${wp}_$b.log 2>&1
    python -c "
import json; d=json.load(open('$OUT/llama8b_k65536_r256_window_parts_${wp}_batch$b.json')); print('VPTQ_SLICED_WINDOW_PARTS=$wp batch $b:', round(d['decode_tok_s_hipgraph'],1), 'tok/s;  VQuantLinear', round(d['vqlinear_us_per_token'],1), 'us per step')" | tee -a $OUT/llama.txt
  done
done
