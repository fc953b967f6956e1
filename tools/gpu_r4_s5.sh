#!/bin/bash
# round 4, GPU call 5: the two-table sliced path (v8-k65536-65536): parity tests + timing against the gather kernel; full suite
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4s5; mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -25 | tee $OUT/gpu_suite.txt
timeout 200 python tools/sliced_bench.py --kr 65536 --shapes "8192,8192;4096,4096;4096,14336;14336,4096;28672,8192" --out $OUT/sliced_k65536_r65536.json 2>&1 | grep -v amdgpu.ids | tee $OUT/sliced_k65536_r65536.txt
