#!/bin/bash
# round 3, session 2: queue depth of the chain kernel (bytes in flight per CU), same box
OUT=gpurun_out/r3b; mkdir -p $OUT
B=$PWD/tools/_build
VPTQ_HIP_LIB=$B/libvptq_hip_lim.so timeout 900 python -m pytest tests/test_chain_gpu.py -x -q -m gpu 2>&1 | tail -8 | tee $OUT/test_chain_lim.txt
timeout 300 python tools/chain_bench.py --hidden 8192 --out $OUT/chain_8192_d4.json 2>&1 | tee $OUT/chain_8192_d4.txt
for v in d2 d3 d6 abl1 abl2 abl3 abl3d6; do
  echo "--- $v"
  VPTQ_HIP_LIB=$B/libvptq_hip_$v.so timeout 300 python tools/chain_bench.py --hidden 8192 --modes t1,chain32 --out $OUT/chain_8192_$v.json 2>&1 | grep -v amdgpu.ids | tee $OUT/chain_8192_$v.txt
done
timeout 300 python tools/chain_bench.py --hidden 4096 --modes single,t1,chain8,chain32,dep --out $OUT/chain_4096.json 2>&1 | grep -v amdgpu.ids | tee $OUT/chain_4096.txt
