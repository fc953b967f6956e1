#!/bin/bash
mkdir -p gpurun_out
timeout 600 python tools/prefill_bench.py --out gpurun_out/prefill.json 2>&1 | grep -E "^\{|Error|Traceback" | cut -c1-400
