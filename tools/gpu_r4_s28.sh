#!/bin/bash
# round 4, step 28: 3 - 4 tokens over the sliced layouts with the contraction on the matrix pipe (transposing gathers): parity, timing
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4s28; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gemv_sliced_gpu.py -m gpu -q -p no:cacheprovider --tb=short -k tokens 2>&1 | tail -25 | tee $OUT/tests.txt
for lib in "" st_nomf; do
  echo "== ${lib:-mfma}" | tee -a $OUT/timing.txt
  for cfg in "8 0" "8 256" "8 65536"; do
    set -- $cfg
    VPTQ_HIP_LIB=${lib:+$R/tools/_build/libvptq_hip_$lib.so} timeout 200 python tools/sliced_tokens_bench.py --v $1 --kr $2 --shapes "8192,8192;4096,4096;14336,4096" --only-one-launch 2>&1 | grep -v amdgpu.ids | tee -a $OUT/timing.txt
  done
done
