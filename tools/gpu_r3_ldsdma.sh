#!/bin/bash
# round 3: the LDS-resident formats with the codebook copied by LDS-DMA, against the copy through registers (same box)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r3lds; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "lds or v2 or other_formats" 2>&1 | tail -4 | tee $OUT/tests.txt
for v in "" nodma ""; do
lib=""; [ -n "$v" ] && lib=$R/tools/_build/libvptq_hip_$v.so
echo "== ${v:-dma}" | tee -a $OUT/formats_lds.txt
VPTQ_HIP_LIB=$lib timeout 300 python tools/format_bench.py --hidden 8192 --formats v8-k8192-256,v8-k4096-256,v8-k4096-0,v8-k8192-0,v8-k1024-256 2>&1 | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('%-18s T=%2d  %-22s %7.2f us %6.0f GB/s | diff %.1e' % (d['format'], d['T'], d['default']['kernel'], d['default']['us_per_launch'], d['default']['GBps'], d['max_rel_diff_default_vs_generic']))" | tee -a $OUT/formats_lds.txt
VPTQ_HIP_LIB=$lib VPTQ_EXACT=1 timeout 300 python tools/format_bench.py --hidden 8192 --formats v8-k8192-256 2>&1 | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('exact %-18s %-22s %7.2f us' % (d['format'], d['default']['kernel'], d['default']['us_per_launch']))" | tee -a $OUT/formats_lds.txt
done
