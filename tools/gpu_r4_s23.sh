#!/bin/bash
# round 4, step 23: element blocks in flight per wave in the 2 - 4 token sliced kernel (4 / 8 / 16)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4s23; mkdir -p $OUT
cd $R
for lib in st_q4 "" st_q16; do
  echo "== ${lib:-q8}" | tee -a $OUT/queue.txt
  for cfg in "8 0" "8 256" "8 65536"; do
    set -- $cfg
    VPTQ_HIP_LIB=${lib:+$R/tools/_build/libvptq_hip_$lib.so} timeout 200 python tools/sliced_tokens_bench.py --v $1 --kr $2 --shapes "8192,8192;4096,4096" --only-one-launch 2>&1 | grep -v amdgpu.ids | tee -a $OUT/queue.txt
  done
done
