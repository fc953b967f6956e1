#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r3sl; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gemv_sliced_gpu.py -x -q 2>&1 | tail -5 | tee $OUT/tests.txt
timeout 600 python tools/sliced_bench.py --ring 6 --out $OUT/sliced_bench.json 2>&1 | grep -v amdgpu.ids | tee $OUT/sliced_bench.txt
for v in slq16 slq24; do
echo "== $v" | tee -a $OUT/sliced_bench.txt
VPTQ_HIP_LIB=$R/tools/_build/libvptq_hip_$v.so timeout 600 python tools/sliced_bench.py --ring 6 --shapes "8192,8192" 2>&1 | grep -v amdgpu.ids | cut -c1-240 | tee -a $OUT/sliced_bench.txt
done
timeout 600 python tools/sliced_bench.py --ring 6 --shapes "8192,8192" --bf16 2>&1 | grep -v amdgpu.ids | cut -c1-240 | tee -a $OUT/sliced_bench.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o sb -- python $R/tools/sliced_bench.py --shapes 8192,8192 --ring 4 > /dev/null 2>&1
grep -E "vptq::" $OUT/stats/sb_kernel_stats.csv | sed 's/(.*)",/",/' | cut -c1-120
rm -f $OUT/stats/sb_kernel_trace.csv $OUT/stats/sb_agent_info.csv
