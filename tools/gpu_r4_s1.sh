#!/bin/bash
# round 4, GPU call 1: energy table (every mode of tools/ubench_energy.hip), GPU suite of the round-3 build, counting fuzz of the
# default arithmetic (default build and the f16(c + r)-first build), the plain bench line of this box
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4s1; mkdir -p $OUT
cd $R
timeout 120 tools/_build/ubench_energy 1.5 2>&1 | tee $OUT/ubench_energy_table.txt
timeout 150 python -m pytest tests -q -m gpu -x 2>&1 | tail -3 | tee $OUT/gpu_suite.txt
timeout 300 python tools/gpu_fuzz_count.py --layers 2048 --dtype f16 --chain 32 2>&1 | grep -v amdgpu.ids | tee $OUT/fuzz_count_f16.txt
timeout 300 python tools/gpu_fuzz_count.py --layers 2048 --dtype bf16 --chain 32 --seed 1 2>&1 | grep -v amdgpu.ids | tee $OUT/fuzz_count_bf16.txt
VPTQ_HIP_LIB=$R/tools/_build/libvptq_hip_preadd.so timeout 200 python tools/gpu_fuzz_count.py --layers 1024 --dtype f16 --chain 32 --spot 0 2>&1 | grep -v amdgpu.ids | tee $OUT/fuzz_count_f16_preadd.txt
timeout 150 python bench.py 2> $OUT/bench_stderr.txt | tail -1 > $OUT/bench_default.json
cut -c1-400 $OUT/bench_default.json
