#!/bin/bash
# round 3, last session: fuzz of the kernels whose end-of-stream handling / pre-pass changed (chain launch, batched decode)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r3fuzz2; mkdir -p $OUT
cd $R
timeout 600 python tools/gpu_fuzz.py --tokens --cases 30 --seed 11 2>&1 | grep -v amdgpu.ids | tail -32 | tee $OUT/fuzz_tokens_f16.txt
timeout 600 python tools/gpu_fuzz.py --tokens --cases 20 --seed 12 --dtype bf16 2>&1 | grep -v amdgpu.ids | tail -22 | tee $OUT/fuzz_tokens_bf16.txt
timeout 600 python tools/gpu_fuzz.py --chains --cases 25 --seed 13 2>&1 | grep -v amdgpu.ids | tail -27 | tee $OUT/fuzz_chains_f16.txt
timeout 600 python tools/gpu_fuzz.py --chains --cases 15 --seed 14 --dtype bf16 2>&1 | grep -v amdgpu.ids | tail -17 | tee $OUT/fuzz_chains_bf16.txt
