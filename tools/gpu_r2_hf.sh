#!/bin/bash
OUT=gpurun_out/r2j; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q -s -k "hf_from_pretrained or fused" 2>&1 | tail -12 | tee $OUT/tests_hf.txt
timeout 120 tools/_build/ubench_stream 512 8192 2>&1 | tee $OUT/ubench_stream_4096.txt
