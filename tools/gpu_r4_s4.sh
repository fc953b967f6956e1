#!/bin/bash
# round 4, GPU call 4: the whole GPU suite (all failures listed), f16(c + r)-first build against the default build with
# package power and clock (chain of 32), hidden 4096 single launches with / without the read-ahead hint
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4s4; mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -40 | tee $OUT/gpu_suite.txt
for rep in 1 2; do
  timeout 100 python tools/chain_bench.py --hidden 8192 --modes chain32 --soak 2 2>&1 | grep -v amdgpu.ids | sed "s/^/default rep$rep /" | tee -a $OUT/chain_preadd_ab_power.txt
  VPTQ_HIP_LIB=$R/tools/_build/libvptq_hip_preadd.so timeout 100 python tools/chain_bench.py --hidden 8192 --modes chain32 --soak 2 2>&1 | grep -v amdgpu.ids | sed "s/^/preadd  rep$rep /" | tee -a $OUT/chain_preadd_ab_power.txt
done
timeout 100 python bench.py --hidden 4096 --mode single --no-extras --no-cpu-baseline 2> /dev/null | tail -1 | cut -c1-260 | tee $OUT/h4096_single.txt
timeout 100 python bench.py --hidden 4096 --mode single --no-extras --no-cpu-baseline --prefetch 2> /dev/null | tail -1 | cut -c1-260 | tee $OUT/h4096_single_prefetch.txt
