#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r5s13; mkdir -p $OUT
cd $R
ab() {
  echo "== $1" >> $OUT/ab.txt
  env $2 timeout 300 python tools/sliced_bench.py $3 --shapes "8192,8192;4096,4096;14336,4096" 2>&1 | grep -v amdgpu.ids >> $OUT/ab.txt
}
ab "exact RG kr=65536 Q=2" "X=1" "--exact --kr 65536"
ab "exact RG kr=65536 Q=2 nt gathers" "VPTQ_HIP_LIB=$R/tools/_build/libvptq_hip_rgnt.so" "--exact --kr 65536"
ab "exact RG v16 kr=65536 Q=2" "X=1" "--exact --v 16 --kr 65536"
ab "exact RG v16 kr=65536 Q=2 nt gathers" "VPTQ_HIP_LIB=$R/tools/_build/libvptq_hip_rgnt.so" "--exact --v 16 --kr 65536"
ab "exact RG kr=4096 Q=2" "X=1" "--exact --kr 4096"
cat $OUT/ab.txt | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('=='): print(l.strip()); continue
    try: r = json.loads(l)
    except Exception: print(l.strip()[:300]); continue
    print(f\"  {r['I']}x{r['O']} gather {r['default_us']:.2f} sliced {r['sliced_us']:.2f} slices {r['slices']} rel {r['rel_diff']:.1e}\")
"
