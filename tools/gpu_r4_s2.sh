#!/bin/bash
# round 4, GPU call 2: loops with the reference's roundings priced (ubench modes 40 / 41 / 43 / 47), the new GPU tests (reference
# roundings inside the chain launch, 32 distinct 8192^2 layers, sliced-route fall-backs), counting fuzz with the default build and
# the f16(c + r)-first build, chain of 32: folded vs reference roundings with power / clock
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4s2; mkdir -p $OUT
cd $R
timeout 60 tools/_build/ubench_energy 1.5 40 41 43 47 2>&1 | tee $OUT/ubench_energy_loops.txt
timeout 400 python -m pytest tests/test_chain_gpu.py tests/test_gemv_sliced_gpu.py -q -m gpu -x -s 2>&1 | grep -v amdgpu.ids | tail -8 | tee $OUT/gpu_new_tests.txt
timeout 200 python tools/chain_bench.py --hidden 8192 --modes single,singlex,chain32,chain32x --soak 1.5 --out $OUT/chain_8192_exact.json 2>&1 | grep -v amdgpu.ids | tee $OUT/chain_8192_exact.txt
timeout 300 python tools/gpu_fuzz_count.py --layers 2048 --dtype f16 --chain 32 2>&1 | grep -v amdgpu.ids | tee $OUT/fuzz_count_f16.txt
timeout 200 python tools/gpu_fuzz_count.py --layers 1024 --dtype bf16 --chain 32 --seed 1 2>&1 | grep -v amdgpu.ids | tee $OUT/fuzz_count_bf16.txt
VPTQ_HIP_LIB=$R/tools/_build/libvptq_hip_preadd.so timeout 200 python tools/gpu_fuzz_count.py --layers 2048 --dtype f16 --chain 32 --spot 0 2>&1 | grep -v amdgpu.ids | tee $OUT/fuzz_count_f16_preadd.txt
