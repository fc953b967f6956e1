#!/bin/bash
# round 5, session 29 (the round's last GPU minutes): 2 / 3 tokens over COLUMN PARTS (28672 columns: each part in two window parts) - the new
# tests first, then the whole GPU suite, then the Llama-3-70B-shaped decode at 2 / 3 sequences
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r5s29; mkdir -p $OUT
cd $R
timeout 200 python -m pytest tests/test_gemv_sliced_gpu.py -x -q -m gpu -k "column_parts or (reference_roundings and (28672 or 24576 or 16392))" 2>&1 | tail -6 > $OUT/new_tests.txt; tail -3 $OUT/new_tests.txt
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --tb=short 2>&1 | tail -30 > $OUT/gpu_suite.txt
grep -E "^FAILED|passed|failed" $OUT/gpu_suite.txt | tail -8
for b in 2 3; do
  timeout 200 python tools/llama_decode.py --model 70b --layers 16 --fuse --k 65536 --kr 256 --prompt 64 --new 64 --batch $b --out $OUT/llama70b16_k65536_r256_tokens_over_parts_batch$b.json > $OUT/llama_$b.log 2>&1
  python -c "
import json; d=json.load(open('$OUT/llama70b16_k65536_r256_tokens_over_parts_batch$b.json')); print('tokens over column parts, batch $b:', round(d['decode_tok_s_hipgraph'],1), 'tok/s;  VQuantLinear', round(d['vqlinear_us_per_token'],1), 'us per step')" 2>&1 | tail -1 | tee -a $OUT/llama.txt
done
