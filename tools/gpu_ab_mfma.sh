#!/bin/bash
# gpurun -- 'bash tools/gpu_ab_mfma.sh' : parity suite under both kernels, phase trace, A/B
mkdir -p gpurun_out/ab
for K in valu mfma; do
  echo "== tests with VPTQ_K256_KERNEL=$K"
  VPTQ_K256_KERNEL=$K bash tools/gpu_tests.sh 2>&1 | tee gpurun_out/ab/tests_$K.txt
done
for f in "" "--fast" "--hot --fast"; do timeout 200 python tools/trace_k256m.py --hidden 8192 $f 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/ab/trace.txt
for H in ${HS:-8192 4096}; do
  for K in ${KS:-valu mfma}; do
    echo "== H=$H kernel=$K"
    VPTQ_K256_KERNEL=$K timeout 300 python tools/microbench.py --hidden $H --group 4 --out gpurun_out/ab/mb_${H}_${K}.json 2>&1 | grep "^exact\|^fast"
  done
done
for M in ${MODELS:-70b 8b}; do
  for K in valu mfma; do
    for F in 0 1; do
      VPTQ_K256_KERNEL=$K timeout 300 python tools/shape_bench.py --model $M --flags $F --tokens 1 --out gpurun_out/ab/shapes_${M}_${K}_f$F.json 2>&1 | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('  %s flags=%d %5dx%-5d %7.2f us %6.0f GB/s' % (d['kernel'], d['flags'], d['I'], d['O'], d['us_per_launch'], d['GBps']))"
    done
  done
done
