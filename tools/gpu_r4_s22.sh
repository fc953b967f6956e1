#!/bin/bash
# round 4, step 22: what a column phase costs in the 2 - 4 token sliced kernel (phases forced up, timing-only ablations)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4s22; mkdir -p $OUT
cd $R
for lib in "" st_ab4 st_ab6; do
  for ph in 1 2 4; do
    echo "== ${lib:-product} min_phases=$ph" | tee -a $OUT/phases.txt
    VPTQ_SLICED_MIN_PHASES=$ph VPTQ_HIP_LIB=${lib:+$R/tools/_build/libvptq_hip_$lib.so} timeout 200 python tools/sliced_tokens_bench.py --v 8 --kr 0 --shapes "8192,8192;4096,4096" --only-one-launch 2>&1 | grep -v amdgpu.ids | tee -a $OUT/phases.txt
  done
done
