#!/bin/bash
# row parts (VPTQ_K256C_SPLIT = 2 / 4: a row group of 8192 columns = 8 / 16 sweeps, sums half / a quarter as often) and the
# LDS release as a compiler barrier only (VPTQ_K256C_LDS_FENCE=0): correctness with spin-limited builds, then same-box timing
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r3o; mkdir -p $OUT
cd $R
B=$R/tools/_build
for v in sp2lim sp4lim; do
  echo "== $v" | tee -a $OUT/test_chain_split.txt
  VPTQ_HIP_LIB=$B/libvptq_hip_$v.so timeout 600 python -m pytest tests/test_chain_gpu.py -x -q -m gpu 2>&1 | tail -6 | tee -a $OUT/test_chain_split.txt
done
for rep in 1 2; do
  for v in default nofence sp2 sp2nf sp4; do
    L=$B/libvptq_hip_$v.so; [ $v = default ] && L=$R/vptq_amd/libvptq_hip.so
    echo "== $v (round $rep)" | tee -a $OUT/chain_split_ab.txt
    VPTQ_HIP_LIB=$L timeout 300 python tools/chain_bench.py --hidden 8192 --modes chain32,chain4 --reps 3 2>&1 | grep -v amdgpu.ids | tail -2 | tee -a $OUT/chain_split_ab.txt
  done
done
