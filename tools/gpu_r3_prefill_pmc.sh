#!/bin/bash
# round 3: MFMA-pipe-busy of the bf16 prefill shapes bench.py's extras.prefill reports (4096x4096, 14336x4096; 8192 tokens):
# rocprofv3 kernel stats + one PMC pass over tools/prefill_bench.py -> profiles/r03/prefill_bf16_pmc_summary.json
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r3prefill; mkdir -p $OUT
P="python $R/tools/prefill_bench.py --tokens 8192 --shapes 4096,4096;4096,14336 --dtypes bf16"
cd /tmp && export TMPDIR=/tmp
timeout 600 $P --out $OUT/prefill_bf16.json 2>&1 | grep -v amdgpu.ids | tee $OUT/prefill_bf16.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o pf -- $P > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA --output-format csv -d $OUT/pmc -o pf -- $P > /dev/null 2>&1
cd $R; rm -f $OUT/*/pf_kernel_trace.csv $OUT/*/pf_agent_info.csv
python tools/pmc_kernels.py $OUT/pmc $OUT/prefill_bf16_pmc_kernels.json | cut -c1-300
cut -c1-160 $OUT/stats/pf_kernel_stats.csv | head -8
