#!/bin/bash
# round 3: chain kernel with kernarg lookups, peeked counters, designated finaliser, priorities
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r3o; mkdir -p $OUT
cd $R
B=$R/tools/_build
# bounded spins first: a protocol error shows as a failed test, not as a hung GPU
VPTQ_HIP_LIB=$B/libvptq_hip_lim.so timeout 600 python -m pytest tests/test_chain_gpu.py -x -q 2>&1 | tail -5 | tee $OUT/tests_lim.txt
grep -q "passed" $OUT/tests_lim.txt && ! grep -q "failed" $OUT/tests_lim.txt || exit 1
timeout 600 python -m pytest tests/test_chain_gpu.py -x -q 2>&1 | tail -3 | tee $OUT/tests.txt
for v in "" prio0 nobal; do
  lib=""; [ -n "$v" ] && lib=$B/libvptq_hip_$v.so
  echo "== variant '$v'" | tee -a $OUT/chain_bench.txt
  VPTQ_HIP_LIB=$lib timeout 300 python tools/chain_bench.py --modes single,chain32,chain8,dep 2>&1 | grep -v amdgpu.ids | tail -8 | tee -a $OUT/chain_bench.txt
  VPTQ_HIP_LIB=$lib timeout 300 python tools/chain_bench.py --hidden 4096 --modes chain32,dep 2>&1 | grep -v amdgpu.ids | tail -4 | tee -a $OUT/chain_bench.txt
done
timeout 300 python tools/chain_bench.py --hidden 8192 --rows 28672 --ring 8 --modes single,chain8 2>&1 | grep -v amdgpu.ids | tail -3 | tee -a $OUT/chain_bench.txt
timeout 300 python tools/chain_bench.py --bf16 --modes single,chain32 2>&1 | grep -v amdgpu.ids | tail -3 | tee -a $OUT/chain_bench.txt
