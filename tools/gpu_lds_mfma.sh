#!/bin/bash
# gemv_lds_mfma_kernel: parity, then timing against variants: $1 = name=lib,... (default: the kernel with the
# reference's roundings via VPTQ_LDS_KERNEL=valu)
OUT=gpurun_out/r5d; mkdir -p $OUT; rm -f $OUT/formats_lds.txt
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "lds or v2" 2>&1 | tail -15 | tee $OUT/tests.txt
FORMATS=${FORMATS:-v8-k8192-256,v8-k4096-256,v8-k4096-0,v8-k8192-0}
run() {  # name, env assignments...
  local name=$1; shift
  env "$@" timeout 300 python tools/format_bench.py --formats $FORMATS 2>&1 | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$name %-18s T=%2d  %-22s %7.1f us %6.0f GB/s | generic %7.1f us | diff %.1e' % (d['format'], d['T'], d['default']['kernel'], d['default']['us_per_launch'], d['default']['GBps'], d['generic']['us_per_launch'], d['max_rel_diff_default_vs_generic']))" | tee -a $OUT/formats_lds.txt
}
for rep in 1 2; do
  run mfma VPTQ_LDS_KERNEL=mfma
  for kv in ${VARIANTS//,/ }; do run ${kv%%=*} VPTQ_HIP_LIB=${kv#*=}; done
done
run valu VPTQ_LDS_KERNEL=valu
