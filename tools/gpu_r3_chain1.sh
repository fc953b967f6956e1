#!/bin/bash
# round 3, session 1: bring-up of gemv_k256t (transposing gather + layer chain): what the transposing read
# does, parity (bounded-spin build first), then us per layer of every launch structure on one box
OUT=gpurun_out/r3a; mkdir -p $OUT
B=$PWD/tools/_build
$B/tr_probe > $OUT/tr_probe.txt 2>&1; tail -1 $OUT/tr_probe.txt
VPTQ_HIP_LIB=$B/libvptq_hip_lim.so timeout 900 python -m pytest tests/test_chain_gpu.py -x -q -m gpu 2>&1 | tail -25 | tee $OUT/test_chain_lim.txt
if ! grep -q " passed" $OUT/test_chain_lim.txt || grep -q "failed" $OUT/test_chain_lim.txt; then
  echo "--- other transposing-read convention"
  VPTQ_HIP_LIB=$B/libvptq_hip_trv1.so timeout 600 python -m pytest tests/test_chain_gpu.py -x -q -m gpu 2>&1 | tail -25 | tee $OUT/test_chain_trv1.txt
fi
echo "--- default build"
timeout 900 python -m pytest tests/test_chain_gpu.py -q -m gpu 2>&1 | tail -15 | tee $OUT/test_chain.txt
timeout 300 python tools/chain_bench.py --hidden 8192 --out $OUT/chain_8192.json 2>&1 | tee $OUT/chain_8192.txt
for v in roth0 abl1 abl2 abl3; do
  echo "--- $v"
  VPTQ_HIP_LIB=$B/libvptq_hip_$v.so timeout 300 python tools/chain_bench.py --hidden 8192 --modes t1,chain32 --out $OUT/chain_8192_$v.json 2>&1 | tee $OUT/chain_8192_$v.txt
done
timeout 300 python tools/chain_bench.py --hidden 4096 --modes single,t1,chain8,chain32,dep --out $OUT/chain_4096.json 2>&1 | tee $OUT/chain_4096.txt
timeout 300 python tools/chain_bench.py --hidden 8192 --rows 28672 --ring 8 --modes single,t1,chain8 --out $OUT/chain_28672x8192.json 2>&1 | tee $OUT/chain_28672x8192.txt
