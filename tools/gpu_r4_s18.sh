#!/bin/bash
# round 4, step 18: the bias line of the load-time gate at 1 (was 2) + layers with an input permutation inside the chain launch:
# the GPU suite (all failures listed) and the exceedance count again
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4s18; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --tb=short 2>&1 | tail -80 | tee $OUT/suite.txt
timeout 200 python tools/gpu_fuzz_count.py --layers 8192 --dtype f16 --chain 32 --seed 7 --spot 64 2>&1 | grep -v amdgpu.ids | tail -28 | tee $OUT/fuzz_count_f16_8192.txt
timeout 200 python tools/gpu_fuzz_count.py --layers 4096 --dtype bf16 --chain 32 --seed 8 --spot 64 2>&1 | grep -v amdgpu.ids | tail -24 | tee $OUT/fuzz_count_bf16_4096.txt
timeout 300 python tools/gpu_fuzz.py --adversarial --cases 32 --seed 1501 2>&1 | tail -14 | tee $OUT/fuzz_adversarial.txt
timeout 300 python tools/gpu_fuzz.py --chains --cases 25 --seed 1502 2>&1 | tail -1 | tee $OUT/fuzz_chains.txt
