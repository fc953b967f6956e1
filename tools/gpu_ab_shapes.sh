#!/bin/bash
# new prologue order (default build) vs the round-2 order (tools/_build/libvptq_hip_old.so): parity subset, then
# per-shape timings of the Llama-3 projections, interleaved on one box
OUT=gpurun_out/r3i; mkdir -p $OUT
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "mfma or canonical or baseline_size or full_size or grouped or default_and_exact or golden or sibling or shards or determin" 2>&1 | tail -3 | tee $OUT/tests.txt
for rep in 1 2; do
for v in old new; do
  lib=$PWD/tools/_build/libvptq_hip_old.so; [ $v = new ] && lib=$PWD/vptq_amd/libvptq_hip.so
  VPTQ_HIP_LIB=$lib timeout 300 python tools/shape_bench.py --model 70b --tokens 1,2,4 2>&1 | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$v rep$rep %5dx%-5d t%d %7.2f us %6.0f GB/s' % (d['I'], d['O'], d['tokens'], d['us_per_launch'], d['GBps']))" | tee -a $OUT/shapes.txt
done; done
for v in old new; do
  lib=$PWD/tools/_build/libvptq_hip_old.so; [ $v = new ] && lib=$PWD/vptq_amd/libvptq_hip.so
  VPTQ_HIP_LIB=$lib timeout 300 python tools/shape_bench.py --model 8b --tokens 1 2>&1 | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$v %5dx%-5d t%d %7.2f us %6.0f GB/s' % (d['I'], d['O'], d['tokens'], d['us_per_launch'], d['GBps']))" | tee -a $OUT/shapes.txt
done
