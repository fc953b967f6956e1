#!/bin/bash
# round 2, first GPU session: parity suite, streaming floor, phase trace, k256m A/B variants
OUT=gpurun_out/r2a; mkdir -p $OUT
echo "== tests"; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $OUT/tests.txt
echo "== stream floor"; timeout 120 tools/_build/ubench_stream 1024 16384 2>&1 | tee $OUT/ubench_stream_8192.txt
timeout 120 tools/_build/ubench_stream 512 8192 2>&1 | tee $OUT/ubench_stream_4096.txt
timeout 120 tools/_build/ubench_stream 3584 16384 2>&1 | tee $OUT/ubench_stream_28672x8192.txt
echo "== trace"; for f in "" "--hot"; do timeout 200 python tools/trace_k256m.py --hidden 8192 $f 2>&1 | grep -v amdgpu.ids; done | tee $OUT/trace.txt
echo "== A/B"
for rep in 1 2; do
  for v in base acc4 nt acc4nt addform ahead3; do
    lib=$PWD/tools/_build/libvptq_hip_$v.so; [ $v = base ] && lib=$PWD/vptq_amd/libvptq_hip.so
    echo "-- $v rep $rep"
    VPTQ_HIP_LIB=$lib timeout 200 python tools/microbench.py --hidden 8192 --group 4 --variants default --no-copy 2>&1 | grep "^default" | tee -a $OUT/ab_$v.txt
  done
done
echo "== bench"; timeout 600 python bench.py 2>&1 | tail -1 | tee $OUT/bench.json
