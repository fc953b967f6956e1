#!/bin/bash
# round 4, step 24: what a phase switch costs: no staging of later phases (ab8), no barriers either (ab24)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4s24; mkdir -p $OUT
cd $R
for lib in "" st_ab8 st_ab24; do
  for ph in 2 4; do
    echo "== ${lib:-product} min_phases=$ph" | tee -a $OUT/switch.txt
    VPTQ_SLICED_MIN_PHASES=$ph VPTQ_HIP_LIB=${lib:+$R/tools/_build/libvptq_hip_$lib.so} timeout 200 python tools/sliced_tokens_bench.py --v 8 --kr 0 --shapes "8192,8192;4096,4096" --only-one-launch 2>&1 | grep -v amdgpu.ids | tee -a $OUT/switch.txt
  done
done
