#!/bin/bash
# round 3: the final evidence pass (profiles/r03): GPU suite, smoke, bench line + rocprofv3 stats / PMC, sliced-layout table,
# Llama-3-8B shaped decode in both formats
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r3fin; mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -5 | tee $OUT/gpu_suite.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $OUT/smoke.txt
bash tools/gpu_r3_bench1.sh > $OUT/bench1.log 2>&1; tail -3 $OUT/bench1.log | cut -c1-300
timeout 600 python tools/sliced_bench.py --ring 6 --kr 0 --shapes "8192,8192;4096,4096;4096,14336;14336,4096;28672,8192" --out $OUT/sliced_k65536_r0.json 2>&1 | grep -v amdgpu.ids | cut -c1-200 | tee $OUT/sliced_final.txt
timeout 600 python tools/sliced_bench.py --ring 6 --kr 256 --shapes "8192,8192;4096,4096;4096,14336;14336,4096;28672,8192" --out $OUT/sliced_k65536_r256.json 2>&1 | grep -v amdgpu.ids | cut -c1-200 | tee -a $OUT/sliced_final.txt
timeout 600 python tools/sliced_bench.py --ring 6 --kr 256 --shapes "8192,8192" --bf16 --out $OUT/sliced_k65536_r256_bf16.json 2>&1 | grep -v amdgpu.ids | cut -c1-200 | tee -a $OUT/sliced_final.txt
timeout 900 python tools/llama_decode.py --k 65536 --kr 256 --new 64 --out $OUT/llama8b_k65536_r256_sliced.json 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-400 | tee $OUT/llama.txt
VPTQ_SLICED_LAYOUT=0 timeout 900 python tools/llama_decode.py --k 65536 --kr 256 --new 64 --out $OUT/llama8b_k65536_r256_default.json 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-400 | tee -a $OUT/llama.txt
