#!/bin/bash
# round 3: the one-pass batched-decode kernel (gemm_k256t): parity, then us per layer against the round-2 kernels
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r3bt; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gemm_k256t_gpu.py -x -q 2>&1 | tail -15 | tee $OUT/tests.txt
timeout 300 python tools/tokens_bench.py --shapes "8192,8192;4096,4096" --tokens 1,2,3,4,5,8,12,16 2>&1 | grep -v amdgpu.ids | tee $OUT/tokens_ws.txt
timeout 300 python tools/tokens_bench.py --shapes "8192,8192;4096,4096" --tokens 2,4,5,8,16 --no-ws 2>&1 | grep -v amdgpu.ids | tee $OUT/tokens_nows.txt
timeout 300 python tools/tokens_bench.py --shapes "8192,8192" --tokens 2,5,16 --bf16 2>&1 | grep -v amdgpu.ids | tee $OUT/tokens_bf16.txt
