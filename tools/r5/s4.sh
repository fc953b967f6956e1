#!/bin/bash
# round 5, session 4: GPU suite under the reference-default arithmetic; sliced kernel A/Bs (issue priority by wave age,
# 16 instead of 8 slices with the one-hop hand-over)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r5s4; mkdir -p $OUT
cd $R
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --tb=short 2>&1 | tail -80 > $OUT/suite.txt
grep -E "^FAILED|passed|failed" $OUT/suite.txt | tail -40
ab() {  # name, env
  echo "== $1" >> $OUT/sliced_ab.txt
  env $2 timeout 200 python tools/sliced_bench.py --kr $3 --shapes "8192,8192;4096,4096;4096,14336" 2>&1 | grep -v amdgpu.ids >> $OUT/sliced_ab.txt
}
for kr in 0 256; do
  ab "kr=$kr default" "X=1" $kr
  ab "kr=$kr pr1 (youngest first)" "VPTQ_HIP_LIB=$R/tools/_build/libvptq_hip_pr1.so" $kr
  ab "kr=$kr pr2 (oldest first)" "VPTQ_HIP_LIB=$R/tools/_build/libvptq_hip_pr2.so" $kr
  ab "kr=$kr 16 slices" "VPTQ_SLICED_SLICES=16" $kr
done
cat $OUT/sliced_ab.txt | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('=='): print(l.strip()); continue
    try: r = json.loads(l)
    except Exception: continue
    print(f\"  {r['I']}x{r['O']} default {r['default_us']:.2f} sliced {r['sliced_us']:.2f} slices {r['slices']} rpw {r['rows_per_wave']} rel {r['rel_diff']:.1e}\")
"
VPTQ_HIP_LIB=$R/tools/_build/libvptq_hip_pr1.so timeout 120 python tools/sliced_trace.py --kr 0 --raw $OUT/trace_pr1_kr0.npy 2>&1 | grep -v amdgpu.ids > $OUT/trace_pr1_kr0.json
python -c "
import json, numpy as np; d=json.load(open('$OUT/trace_pr1_kr0.json')); print('trace pr1 kr0', d['runs'][-1])
us=np.load('$OUT/trace_pr1_kr0.npy'); print('stream_len by wave', np.round((us[:,:,2]-us[:,:,1]).mean(0),2))"
