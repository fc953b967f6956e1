#!/bin/bash
# gpurun -- 'bash tools/gpu_ab_mfma.sh' : parity suite, phase trace, VALU-vs-MFMA kernel A/B
mkdir -p gpurun_out/ab
bash tools/gpu_tests.sh 2>&1 | tee gpurun_out/ab/tests.txt
for f in "" "--fast" "--hot --fast"; do timeout 200 python tools/trace_k256m.py --hidden 8192 $f 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/ab/trace.txt
for H in ${HS:-8192 4096}; do
  for K in ${KS:-valu mfma}; do
    echo "== H=$H kernel=$K"
    VPTQ_K256_KERNEL=$K timeout 300 python tools/microbench.py --hidden $H --group 4 --out gpurun_out/ab/mb_${H}_${K}.json 2>&1 | grep "^exact\|^fast"
  done
done
