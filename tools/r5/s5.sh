#!/bin/bash
# round 5, session 5: the exact sliced kernel (reference roundings over the layouts): parity, timing against the gather kernel
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r5s5; mkdir -p $OUT
cd $R
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gemv_sliced_gpu.py -m gpu -q -p no:cacheprovider --tb=short -x -k "reference_roundings or exact_sliced" 2>&1 | tail -30 > $OUT/exact_tests.txt
tail -12 $OUT/exact_tests.txt
for kr in 0 256; do
  echo "== exact kr=$kr" >> $OUT/sliced_exact.txt
  timeout 300 python tools/sliced_bench.py --exact --kr $kr --shapes "8192,8192;4096,4096;4096,14336;14336,4096;4096,1024" 2>&1 | grep -v amdgpu.ids >> $OUT/sliced_exact.txt
done
cat $OUT/sliced_exact.txt | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('=='): print(l.strip()); continue
    try: r = json.loads(l)
    except Exception: print(l.strip()[:200]); continue
    print(f\"  {r['I']}x{r['O']} gather {r['default_us']:.2f} sliced {r['sliced_us']:.2f} slices {r['slices']} rpw {r['rows_per_wave']} rel {r['rel_diff']:.1e}\")
"
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --tb=short 2>&1 | tail -40 > $OUT/suite.txt
grep -E "^FAILED|passed|failed" $OUT/suite.txt | tail -30
