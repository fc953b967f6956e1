#!/bin/bash
# round 4, step 20: where the time of the 2 - 4 token sliced kernel goes (timing-only ablation builds) + clock / power while it runs
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4s20; mkdir -p $OUT
cd $R
for lib in "" st_ab1 st_ab2 st_ab3 st_ab4; do
  echo "== ${lib:-product}" | tee -a $OUT/ablate.txt
  for cfg in "8 0" "8 256" "8 65536"; do
    set -- $cfg
    VPTQ_HIP_LIB=${lib:+$R/tools/_build/libvptq_hip_$lib.so} timeout 200 python tools/sliced_tokens_bench.py --v $1 --kr $2 --shapes "8192,8192" --only-one-launch 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ablate.txt
  done
done
