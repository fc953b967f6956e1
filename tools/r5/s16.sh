#!/bin/bash
# round 5, session 16: the exact slice rule that leaves room for two tokens (16 slices from ~4000 columns): sliced GPU tests, the
# Llama-3-8B-shaped decode loop in v8-k65536-256 with 1, 2 and 3 sequences - new rule against VPTQ_SLICED_SLICES=8 (the old one)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r5s16; mkdir -p $OUT; rm -f $OUT/*.txt $OUT/*.json
cd $R
timeout 1200 python -m pytest tests/test_gemv_sliced_gpu.py -x -q -m gpu 2>&1 | tail -8 > $OUT/tests.txt; tail -3 $OUT/tests.txt
for s in new 8; do
  for b in 1 2 3; do
    if [ $s = 8 ]; then export VPTQ_SLICED_SLICES=8; else unset VPTQ_SLICED_SLICES; fi
    timeout 400 python tools/llama_decode.py --fuse --k 65536 --kr 256 --new 128 --batch $b --out $OUT/llama8b_k65536_r256_slices_${s}_batch$b.json > $OUT/llama_${s}_$b.log 2>&1
    python -c "
import json; d=json.load(open('$OUT/llama8b_k65536_r256_slices_${s}_batch$b.json')); print('slices rule $s batch $b:', round(d['decode_tok_s_hipgraph'],1), 'tok/s;  VQuantLinear', round(d['vqlinear_us_per_token'],1), 'us per step')" | tee -a $OUT/llama.txt
  done
done
