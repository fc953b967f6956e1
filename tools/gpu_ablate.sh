#!/bin/bash
# timing-only ablations of gemv_k256m_kernel (VPTQ_K256M_ABLATE bits: 1 no MFMA, 2 no gathers, 4 no epilogue,
# 8 no index loads, 16 two MFMAs, 32 one MFMA per index); nc4* = no operand cache, queue depth 4
OUT=gpurun_out/r3e; mkdir -p $OUT
B=tools/_build
L="full=$B/libvptq_hip_old.so,mfma2=$B/libvptq_hip_m2.so,mfma1=$B/libvptq_hip_m1.so,mfma0=$B/libvptq_hip_m0.so,gather0=$B/libvptq_hip_g0.so,m0g0=$B/libvptq_hip_m0g0.so,nc4=$B/libvptq_hip_nc4.so,nc4m0=$B/libvptq_hip_nc4m0.so,nc4m0g0=$B/libvptq_hip_nc4m0g0.so"
timeout 500 python tools/ab_libs.py --no-parity --libs $L --hidden 8192 --reps 4 --out $OUT/ablate_8192.json 2>&1 | tee $OUT/ablate_8192.txt
