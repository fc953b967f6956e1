#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r3bt; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gemm_k256t_gpu.py -x -q 2>&1 | tail -5 | tee $OUT/tests.txt
timeout 300 python tools/tokens_bench.py --shapes "8192,8192;4096,4096" --tokens 2,5,16 2>&1 | grep -v amdgpu.ids | tee $OUT/tokens_ws2.txt
for v in bt4 bt7; do
  echo "== $v" | tee -a $OUT/tokens_abl.txt
  VPTQ_HIP_LIB=$R/tools/_build/libvptq_hip_$v.so timeout 300 python tools/tokens_bench.py --shapes "8192,8192" --tokens 2,16 2>&1 | grep -v amdgpu.ids | tee -a $OUT/tokens_abl.txt
done
cd /tmp && export TMPDIR=/tmp
for T in 16; do
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_t$T -o tb -- python $R/tools/tokens_bench.py --shapes "8192,8192" --tokens $T > /dev/null 2>&1
cut -c1-100 $OUT/stats_t$T/tb_kernel_stats.csv | head -3
grep prep $OUT/stats_t$T/tb_kernel_stats.csv | sed 's/.*)",/prep: /'
rm -f $OUT/stats_t$T/tb_kernel_trace.csv $OUT/stats_t$T/tb_agent_info.csv
done
