#!/bin/bash
# round 3, session 8: gemv_k256c (clean chain kernel: 4x4x4 loop, 2 row subgroups per sweep)
OUT=gpurun_out/r3j; mkdir -p $OUT
B=$PWD/tools/_build
VPTQ_HIP_LIB=$B/libvptq_hip_lim.so timeout 900 python -m pytest tests/test_chain_gpu.py -x -q -m gpu 2>&1 | tail -8 | tee $OUT/test_chain_lim.txt
timeout 600 python -m pytest tests/test_chain_gpu.py -q -m gpu 2>&1 | tail -4 | tee $OUT/test_chain.txt
timeout 300 python tools/chain_bench.py --hidden 8192 --out $OUT/chain_8192.json 2>&1 | grep -v amdgpu.ids | tee $OUT/chain_8192.txt
for v in sub1 d2 d4 ms4 ms16 abl3; do
  echo "--- $v"
  VPTQ_HIP_LIB=$B/libvptq_hip_$v.so timeout 300 python tools/chain_bench.py --hidden 8192 --modes chain32 --out $OUT/chain_8192_$v.json 2>&1 | grep -v amdgpu.ids | tee $OUT/chain_8192_$v.txt
done
echo "--- tall layer 65536 x 8192"
timeout 300 python tools/chain_bench.py --hidden 8192 --rows 65536 --ring 4 --modes single,t1 --out $OUT/tall.json 2>&1 | grep -v amdgpu.ids | tee $OUT/tall.txt
timeout 300 python tools/chain_bench.py --hidden 4096 --modes single,t1,chain8,chain32,dep --out $OUT/chain_4096.json 2>&1 | grep -v amdgpu.ids | tee $OUT/chain_4096.txt
timeout 300 python tools/chain_bench.py --hidden 8192 --rows 28672 --ring 8 --modes single,chain8 --out $OUT/chain_28672x8192.json 2>&1 | grep -v amdgpu.ids | tee $OUT/chain_28672x8192.txt
