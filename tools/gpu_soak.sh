#!/bin/bash
# soak: the GPU suite three times + random-layer fuzzers with fresh seeds (flaky races show up as rare failures)
OUT=gpurun_out/r5o; mkdir -p $OUT; S=${SEED:-900}
for i in 1 2 3; do timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -1; done | tee $OUT/suite3.txt
timeout 600 python tools/gpu_fuzz.py --cases 40 --seed $S 2>&1 | tail -1 | tee $OUT/fuzz_canon.txt
timeout 600 python tools/gpu_fuzz.py --cases 30 --seed $((S+1)) --dtype bf16 2>&1 | tail -1 | tee -a $OUT/fuzz_canon.txt
timeout 600 python tools/gpu_fuzz.py --formats --cases 200 --seed $((S+2)) 2>&1 | tail -1 | tee $OUT/fuzz_formats.txt
timeout 600 python tools/gpu_fuzz.py --formats --cases 100 --seed $((S+3)) --dtype bf16 2>&1 | tail -1 | tee -a $OUT/fuzz_formats.txt
timeout 600 python tools/gpu_fuzz.py --lds-tall --cases 40 --seed $((S+4)) 2>&1 | tail -1 | tee $OUT/fuzz_lds.txt
timeout 600 python tools/gpu_fuzz.py --lds-tall --cases 30 --seed $((S+5)) --dtype bf16 2>&1 | tail -1 | tee -a $OUT/fuzz_lds.txt
