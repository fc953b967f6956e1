#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r3sl; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gemv_sliced_gpu.py -x -q 2>&1 | tail -3 | tee $OUT/tests.txt
for cfg in "1:" "4:" "1:q4" "1:q16"; do
epl=${cfg%%:*}; v=${cfg#*:}
lib=""; [ -n "$v" ] && lib=$R/tools/_build/libvptq_hip_$v.so
echo "== epl $epl queue ${v:-q8}" | tee -a $OUT/sliced_ab3.txt
VPTQ_HIP_LIB=$lib timeout 600 python tools/sliced_bench.py --ring 6 --epl $epl --shapes "8192,8192;4096,4096" 2>&1 | grep -v amdgpu.ids | cut -c1-215 | tee -a $OUT/sliced_ab3.txt
done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o sb -- python $R/tools/sliced_bench.py --shapes 8192,8192 --ring 4 > /dev/null 2>&1
grep -E "vptq::" $OUT/stats/sb_kernel_stats.csv | sed 's/(.*)",/",/' | cut -c1-120
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_INSTS_LDS --output-format csv -d $OUT/pmc -o sb -- python $R/tools/sliced_bench.py --shapes 8192,8192 --ring 2 > /dev/null 2>&1
cd $R; rm -f $OUT/*/sb_kernel_trace.csv $OUT/*/sb_agent_info.csv
python tools/pmc_kernels.py $OUT/pmc $OUT/sliced_pmc_kernels.json sliced_kernel | cut -c1-400
