#!/bin/bash
# round 4, step 26: the 2 - 4 token sliced kernel with counted waits in the stream loop (no compiler-visible load inside it, one
# load per step on every path), permutation by a pre-pass: parity, timing, 8 against 16 blocks in flight
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4s26; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gemv_sliced_gpu.py -m gpu -q -p no:cacheprovider --tb=short -x 2>&1 | tail -15 | tee $OUT/tests.txt
for lib in "" st_q16; do
  echo "== ${lib:-q8}" | tee -a $OUT/queue.txt
  for cfg in "8 0" "8 256" "8 65536" "16 65536"; do
    set -- $cfg
    VPTQ_HIP_LIB=${lib:+$R/tools/_build/libvptq_hip_$lib.so} timeout 200 python tools/sliced_tokens_bench.py --v $1 --kr $2 --shapes "8192,8192;4096,4096;14336,4096" --only-one-launch 2>&1 | grep -v amdgpu.ids | tee -a $OUT/queue.txt
  done
done
