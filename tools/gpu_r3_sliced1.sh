#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r3sl; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gemv_sliced_gpu.py -x -q 2>&1 | tail -15 | tee $OUT/tests.txt
timeout 600 python tools/sliced_bench.py --out $OUT/sliced_bench.json 2>&1 | grep -v amdgpu.ids | tee $OUT/sliced_bench.txt
