#!/bin/bash
# round 5, session 19: Llama-3-70B shapes again (16 of 80 blocks, v8-k65536-256), now with the 28672-column down projections served as
# two column parts by the exact sliced kernel; 1 - 3 sequences
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r5s19; mkdir -p $OUT; rm -f $OUT/*.txt $OUT/*.json
cd $R
for b in 1 2 3; do
  timeout 500 python tools/llama_decode.py --model 70b --layers 16 --fuse --k 65536 --kr 256 --prompt 64 --new 64 --batch $b --out $OUT/llama70b16_k65536_r256_column_parts_batch$b.json > $OUT/llama_$b.log 2>&1
  python -c "
import json; d=json.load(open('$OUT/llama70b16_k65536_r256_column_parts_batch$b.json')); print('column parts, batch $b:', round(d['decode_tok_s_hipgraph'],1), 'tok/s;  VQuantLinear', round(d['vqlinear_us_per_token'],1), 'us per step')" 2>&1 | tail -1 | tee -a $OUT/llama.txt
done
