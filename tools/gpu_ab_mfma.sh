#!/bin/bash
# gpurun -- 'bash tools/gpu_ab_mfma.sh' : parity suite with the kernel choice pinned either way,
# phase traces of both kernels, kernel A/B (flags: default / valu / mfma / exact), Llama shapes
mkdir -p gpurun_out/ab
for K in valu mfma; do
  echo "== tests with VPTQ_K256_KERNEL=$K"
  VPTQ_K256_KERNEL=$K bash tools/gpu_tests.sh 2>&1 | tee gpurun_out/ab/tests_$K.txt
done
for f in "" "--hot" "--exact" "--kernel valu"; do timeout 200 python tools/trace_k256m.py --hidden 8192 $f 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/ab/trace.txt
for H in ${HS:-8192 4096}; do
  timeout 300 python tools/microbench.py --hidden $H --group 4 --out gpurun_out/ab/kernel_ab_h$H.json 2>&1 | grep -v "^{"
done
for M in ${MODELS:-70b 8b}; do
  for K in valu mfma; do
    VPTQ_K256_KERNEL=$K timeout 300 python tools/shape_bench.py --model $M --tokens 1 --out gpurun_out/ab/shapes_${M}_$K.json 2>&1 | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('  %s %5dx%-5d %7.2f us %6.0f GB/s' % (d['kernel'], d['I'], d['O'], d['us_per_launch'], d['GBps']))"
  done
done
