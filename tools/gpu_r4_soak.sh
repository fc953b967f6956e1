#!/bin/bash
# round 4 soak: the GPU suite twice more + every fuzzer with fresh seeds on the final build (flaky races show up as rare failures)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4soak; mkdir -p $OUT; S=${SEED:-1400}
cd $R
for i in 1 2; do timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -1; done | tee $OUT/suite2.txt
timeout 300 python tools/gpu_fuzz.py --cases 40 --seed $S 2>&1 | tail -1 | tee $OUT/fuzz_canon.txt
timeout 300 python tools/gpu_fuzz.py --cases 30 --seed $((S+1)) --dtype bf16 2>&1 | tail -1 | tee -a $OUT/fuzz_canon.txt
timeout 400 python tools/gpu_fuzz.py --formats --cases 200 --seed $((S+2)) 2>&1 | tail -1 | tee $OUT/fuzz_formats.txt
timeout 300 python tools/gpu_fuzz.py --formats --cases 100 --seed $((S+3)) --dtype bf16 2>&1 | tail -1 | tee -a $OUT/fuzz_formats.txt
timeout 300 python tools/gpu_fuzz.py --lds-tall --cases 30 --seed $((S+4)) 2>&1 | tail -1 | tee $OUT/fuzz_lds.txt
timeout 300 python tools/gpu_fuzz.py --chains --cases 25 --seed $((S+5)) 2>&1 | tail -1 | tee $OUT/fuzz_chains.txt
timeout 300 python tools/gpu_fuzz.py --chains --cases 15 --seed $((S+6)) --dtype bf16 2>&1 | tail -1 | tee -a $OUT/fuzz_chains.txt
timeout 300 python tools/gpu_fuzz.py --tokens --cases 30 --seed $((S+7)) 2>&1 | tail -1 | tee $OUT/fuzz_tokens.txt
timeout 300 python tools/gpu_fuzz.py --adversarial --cases 32 --seed $((S+8)) 2>&1 | tail -14 | tee $OUT/fuzz_adversarial.txt
timeout 300 python tools/gpu_fuzz.py --sliced --cases 60 --seed $((S+9)) 2>&1 | tail -1 | tee $OUT/fuzz_sliced.txt
timeout 200 python tools/gpu_fuzz_count.py --layers 8192 --dtype f16 --chain 32 --seed 7 --spot 64 2>&1 | grep -v amdgpu.ids | tail -26 | tee $OUT/fuzz_count_f16_8192.txt
timeout 200 python tools/gpu_fuzz_count.py --layers 4096 --dtype bf16 --chain 32 --seed 8 --spot 64 2>&1 | grep -v amdgpu.ids | tail -22 | tee $OUT/fuzz_count_bf16_4096.txt
