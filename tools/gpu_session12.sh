#!/bin/bash
mkdir -p gpurun_out
timeout 300 python -u tools/llama_decode.py --layers 2 --prompt 16 --new 8 2>&1 | grep -v Warning | tail -3 | cut -c1-600
timeout 900 python -u tools/llama_decode.py --layers 32 --out gpurun_out/llama8b_decode.json 2>&1 | tail -1 | cut -c1-900
timeout 900 python -u tools/llama_decode.py --layers 32 --perm --out gpurun_out/llama8b_decode_perm.json 2>&1 | tail -1 | cut -c1-900
