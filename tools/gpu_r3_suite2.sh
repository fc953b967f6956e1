#!/bin/bash
# round 3: the whole GPU suite with the final library + the token tables of the batched-decode routes
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r3suite2; mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -12 | tee $OUT/gpu_suite.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $OUT/smoke.txt
timeout 300 python tools/tokens_bench.py --shapes "8192,8192;4096,4096;8192,28672" --tokens 1,2,4,5,8,16 --bf16 --out $OUT/tokens_bf16.json 2>&1 | grep -v amdgpu.ids | tee $OUT/tokens_bf16.txt
VPTQ_GEMMT_MIN_TOKENS_BF16=99 timeout 300 python tools/tokens_bench.py --shapes "8192,8192;4096,4096" --tokens 5,8,16 --bf16 --out $OUT/tokens_bf16_round2_route.json 2>&1 | grep -v amdgpu.ids | tee $OUT/tokens_bf16_round2_route.txt
timeout 300 python tools/tokens_bench.py --shapes "8192,8192;4096,4096;8192,28672" --tokens 1,2,4,5,8,16 --out $OUT/tokens_f16.json 2>&1 | grep -v amdgpu.ids | tee $OUT/tokens_f16.txt
VPTQ_GEMMT_MIN_TOKENS_F16=99 timeout 300 python tools/tokens_bench.py --shapes "8192,8192;4096,4096;8192,28672" --tokens 5,8,16 --out $OUT/tokens_f16_round2_route.json 2>&1 | grep -v amdgpu.ids | tee $OUT/tokens_f16_round2_route.txt
VPTQ_GEMMT_MIN_TOKENS_F16=2 timeout 300 python tools/tokens_bench.py --shapes "8192,8192;4096,4096" --tokens 2,3,4 --out $OUT/tokens_f16_k256t_2to4.json 2>&1 | grep -v amdgpu.ids | tee $OUT/tokens_f16_k256t_2to4.txt
