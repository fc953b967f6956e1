#!/bin/bash
# 8 waves per workgroup (2 per SIMD, 256 registers) x 4 row subgroups x deeper gather lookahead against the default
# (16 waves x 2 subgroups x lookahead 2): correctness with a spin-limited build first, then same-box interleaved timing
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r3o; mkdir -p $OUT
cd $R
B=$R/tools/_build
VPTQ_HIP_LIB=$B/libvptq_hip_w8lim.so timeout 600 python -m pytest tests/test_chain_gpu.py -x -q -m gpu 2>&1 | tail -8 | tee $OUT/test_chain_w8lim.txt
for rep in 1 2; do
  for v in default w8s4a4 w8s4a6 w8s2a4 w8s4a3; do
    L=$B/libvptq_hip_$v.so; [ $v = default ] && L=$R/vptq_amd/libvptq_hip.so
    echo "== $v (round $rep)" | tee -a $OUT/chain_w8_ab.txt
    VPTQ_HIP_LIB=$L timeout 300 python tools/chain_bench.py --hidden 8192 --modes single,chain32 --reps 3 2>&1 | grep -v amdgpu.ids | tail -4 | tee -a $OUT/chain_w8_ab.txt
  done
done
