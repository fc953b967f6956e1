#!/bin/bash
OUT=gpurun_out/r3z6; mkdir -p $OUT
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "dequant or golden or baseline_size or token_counts or loader or hf_from" 2>&1 | tail -3 | tee $OUT/tests.txt
for v in prev new; do
  lib=$PWD/tools/_build/libvptq_hip_$v.so; [ $v = new ] && lib=$PWD/vptq_amd/libvptq_hip.so
  VPTQ_HIP_LIB=$lib timeout 300 python tools/prefill_bench.py --tokens 2048 --shapes "4096,4096;8192,8192" --dtypes f16,bf16 2>&1 | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$v %s %5dx%-5d t%d dequant %6.1f us  dense %7.1f us  fused %7.1f us' % (d['dtype'], d['I'], d['O'], d['tokens'], d['dequant_us'], d['dense_us'], d['fused_us']))" | tee -a $OUT/dequant.txt
done
