#!/bin/bash
mkdir -p gpurun_out
timeout 900 python tools/llama_decode.py --layers 32 --out gpurun_out/llama8b_decode.json 2>&1 | tail -4 | cut -c1-900
timeout 900 python tools/llama_decode.py --layers 32 --perm --out gpurun_out/llama8b_decode_perm.json 2>&1 | tail -2 | cut -c1-900
