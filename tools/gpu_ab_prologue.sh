#!/bin/bash
OUT=gpurun_out/r3h; mkdir -p $OUT
B=tools/_build
hostname | tee $OUT/box.txt; rocm-smi --showclocks --showpower --showperflevel 2>/dev/null | grep -v "^=\|^$" | head -20 | tee -a $OUT/box.txt
L="old=$B/libvptq_hip_old.so,dp0s1=$B/libvptq_hip_dp0s1.so,lb1=$B/libvptq_hip_lb1.so,lb2=$B/libvptq_hip_lb2.so,map=$B/libvptq_hip_map.so,maplb1=$B/libvptq_hip_maplb1.so,af=$B/libvptq_hip_af.so"
timeout 500 python tools/ab_libs.py --libs $L --hidden 8192 --reps 4 --out $OUT/ab_8192.json 2>&1 | tee $OUT/ab_8192.txt
rocm-smi --showclocks --showpower 2>/dev/null | grep -v "^=\|^$" | head -12 | tee -a $OUT/box.txt
