#!/bin/bash
# round 5, session 12: cheaper row end (fixed-point conversion) + scalar-base stream loads: parity + timing; RG queue depth A/B
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r5s12; mkdir -p $OUT
cd $R
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gemv_sliced_gpu.py -m gpu -q -p no:cacheprovider --tb=short -x 2>&1 | tail -4
ab() {
  echo "== $1" >> $OUT/ab.txt
  env $2 timeout 300 python tools/sliced_bench.py $3 --shapes "8192,8192;4096,4096;14336,4096" 2>&1 | grep -v amdgpu.ids >> $OUT/ab.txt
}
ab "exact kr=0" "X=1" "--exact --kr 0"
ab "exact kr=256" "X=1" "--exact --kr 256"
ab "folded kr=0" "X=1" "--kr 0"
ab "folded kr=256" "X=1" "--kr 256"
ab "folded kr=65536" "X=1" "--kr 65536"
ab "exact RG kr=65536 Q=8" "X=1" "--exact --kr 65536"
ab "exact RG kr=65536 Q=4" "VPTQ_HIP_LIB=$R/tools/_build/libvptq_hip_rgq4.so" "--exact --kr 65536"
ab "exact RG kr=65536 Q=2" "VPTQ_HIP_LIB=$R/tools/_build/libvptq_hip_rgq2.so" "--exact --kr 65536"
ab "exact RG v16 kr=65536 Q=4" "X=1" "--exact --v 16 --kr 65536"
ab "exact RG v16 kr=65536 Q=2" "VPTQ_HIP_LIB=$R/tools/_build/libvptq_hip_rgq2.so" "--exact --v 16 --kr 65536"
cat $OUT/ab.txt | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('=='): print(l.strip()); continue
    try: r = json.loads(l)
    except Exception: print(l.strip()[:300]); continue
    print(f\"  {r['I']}x{r['O']} gather {r['default_us']:.2f} sliced {r['sliced_us']:.2f} slices {r['slices']} rel {r['rel_diff']:.1e}\")
"
