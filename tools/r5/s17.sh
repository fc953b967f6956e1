#!/bin/bash
# round 5, session 17: Llama-3-70B SHAPES (hidden 8192, 16 of the 80 blocks), v8-k65536-256, default arithmetic, 1 - 3 sequences decoded
# together: the sliced routes (one token: gemv_sliced<EX>; 2 / 3 tokens: one pass, TOK) against the gather kernels alone (VPTQ_SLICED_LAYOUT=0)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r5s17; mkdir -p $OUT; rm -f $OUT/*.txt $OUT/*.json
cd $R
for mode in auto 0; do
  for b in 1 2 3; do
    VPTQ_SLICED_LAYOUT=$mode timeout 500 python tools/llama_decode.py --model 70b --layers 16 --fuse --k 65536 --kr 256 --prompt 64 --new 64 --batch $b --out $OUT/llama70b16_k65536_r256_sliced_${mode}_batch$b.json > $OUT/llama_${mode}_$b.log 2>&1
    python -c "
import json; d=json.load(open('$OUT/llama70b16_k65536_r256_sliced_${mode}_batch$b.json')); print('VPTQ_SLICED_LAYOUT=$mode batch $b:', round(d['decode_tok_s_hipgraph'],1), 'tok/s;  VQuantLinear', round(d['vqlinear_us_per_token'],1), 'us per step')" 2>&1 | tail -1 | tee -a $OUT/llama.txt
  done
done
