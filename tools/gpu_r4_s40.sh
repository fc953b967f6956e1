#!/bin/bash
# round 4, step 40: Llama-3-8B-shaped decode, 4 and 2 sequences at a time, large-codebook formats: every VQuantLinear call sees 4 / 2
# tokens - gather kernels (VPTQ_SLICED_ONE_LAUNCH=0) against one launch over the sliced layouts (auto), siblings grouped
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4s40; mkdir -p $OUT
cd $R
for kr in 256 65536; do
  for b in 4 2; do
    for mode in 0 auto; do
      VPTQ_SLICED_TOKENS=1,1 VPTQ_SLICED_ONE_LAUNCH=$mode timeout 280 python tools/llama_decode.py --fuse --k 65536 --kr $kr --batch $b --new 64 2> /dev/null | tail -1 > $OUT/llama8b_k65536_r${kr}_batch${b}_one_launch_${mode}.json
      python -c "
import json,sys; d=json.load(open('$OUT/llama8b_k65536_r${kr}_batch${b}_one_launch_${mode}.json')); print('kr=$kr batch=$b one_launch=$mode', {k: (round(v,1) if isinstance(v,float) else v) for k,v in d.items() if k in ('decode_tok_s_hipgraph','decode_tok_s_eager','vqlinear_us_per_token','ttft_ms')})"
    done
  done
done 2>&1 | tee $OUT/summary.txt
