#!/bin/bash
OUT=gpurun_out/r3k; mkdir -p $OUT
N=vptq_amd/libvptq_hip.so; O=tools/_build/libvptq_hip_old.so
for H in 4096 5120; do
timeout 300 python tools/ab_libs.py --libs "old=$O,new=$N,old_mfma=$O@8,new_mfma=$N@8,new_valu=$N@16" --hidden $H --reps 4 --group 4 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_$H.txt
done
