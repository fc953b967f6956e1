#!/bin/bash
OUT=gpurun_out/r2l; mkdir -p $OUT
for v in base gnt1 gnt2; do
  lib=$PWD/tools/_build/libvptq_hip_$v.so; [ $v = base ] && lib=$PWD/vptq_amd/libvptq_hip.so
  for kr in 0 256; do
    VPTQ_HIP_LIB=$lib timeout 200 python tools/microbench.py --hidden 8192 --k 65536 --kr $kr --variants default --no-copy 2>&1 | grep -E "^default " | sed "s/^/$v kr=$kr /" | tee -a $OUT/ab_gather.txt
  done
done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 -L 2>/dev/null | grep -oE "\b(TCP|TCC)_[A-Z_0-9]+(_sum)?\b" | sort -u | tr '\n' ' ' | head -c 3000 > $R/$OUT/counters.txt
CMD="python $R/tools/microbench.py --hidden 8192 --k 65536 --kr 0 --variants default --no-copy --iters 3"
timeout 300 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --output-format csv -d $R/$OUT/pmc_a -o g -- $CMD > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_REQ_sum GRBM_GUI_ACTIVE --output-format csv -d $R/$OUT/pmc_b -o g -- $CMD > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $R/$OUT/pmc_c -o g -- $CMD > /dev/null 2>&1
cd $R; rm -f $OUT/*/g_kernel_trace.csv $OUT/*/g_agent_info.csv
python tools/pmc_kernels.py $OUT $OUT/gather_pmc_summary.json gemv_gather
head -c 1500 $OUT/counters.txt
