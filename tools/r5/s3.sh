#!/bin/bash
# round 5, session 3: new sliced prologue order (+ packed add for the 256-entry residual table) against the round-4 build;
# phase stamps; the GPU suite under the new default arithmetic (reference roundings)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r5s3; mkdir -p $OUT
cd $R
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --tb=line 2>&1 | tail -60 > $OUT/suite.txt
tail -45 $OUT/suite.txt
for kr in 0 256 65536; do
  for lib in default epi0; do
    L=""; [ $lib != default ] && L=$R/tools/_build/libvptq_hip_$lib.so
    echo "== kr=$kr lib=$lib" >> $OUT/sliced_ab.txt
    VPTQ_HIP_LIB=$L timeout 200 python tools/sliced_bench.py --kr $kr --shapes "8192,8192;4096,4096;4096,14336;14336,4096;8192,28672" 2>&1 | grep -v amdgpu.ids >> $OUT/sliced_ab.txt
  done
done
cat $OUT/sliced_ab.txt | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('=='): print(l.strip()); continue
    try: r = json.loads(l)
    except Exception: continue
    print(f\"  {r['I']}x{r['O']} default {r['default_us']:.2f} sliced {r['sliced_us']:.2f} rel {r['rel_diff']:.1e}\")
"
for kr in 0 256; do
  VPTQ_HIP_LIB=$R/tools/_build/libvptq_hip_tr1.so timeout 120 python tools/sliced_trace.py --kr $kr --raw $OUT/trace_kr$kr.npy 2>&1 | grep -v amdgpu.ids > $OUT/trace_kr$kr.json
  python -c "
import json; d=json.load(open('$OUT/trace_kr$kr.json')); print('trace kr$kr', d['runs'][-1])"
done
