#!/bin/bash
# round 5, session 8: RG - the reference's roundings for layers with any residual codebook over ONE sliced layout (main entry
# from the LDS slice, residual entry gathered from L2): parity, timing against the gather kernels
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r5s8; mkdir -p $OUT
cd $R
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gemv_sliced_gpu.py -m gpu -q -p no:cacheprovider --tb=short -x 2>&1 | tail -25 > $OUT/sliced_tests.txt
tail -8 $OUT/sliced_tests.txt
ab() {
  echo "== $1" >> $OUT/sliced_rg.txt
  timeout 300 python tools/sliced_bench.py --exact $2 --shapes "8192,8192;4096,4096;4096,14336;14336,4096" 2>&1 | grep -v amdgpu.ids >> $OUT/sliced_rg.txt
}
ab "exact v8 kr=65536" "--kr 65536"
ab "exact v16 kr=65536" "--v 16 --kr 65536"
ab "exact v8 kr=4096" "--kr 4096"
ab "exact v16 kr=1024" "--v 16 --kr 1024"
ab "exact v16 kr=0" "--v 16 --kr 0"
cat $OUT/sliced_rg.txt | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('=='): print(l.strip()); continue
    try: r = json.loads(l)
    except Exception: print(l.strip()[:300]); continue
    print(f\"  {r['I']}x{r['O']} gather {r['default_us']:.2f} ({r['default_kernel']}) sliced {r['sliced_us']:.2f} slices {r['slices']} rel {r['rel_diff']:.1e}\")
"
