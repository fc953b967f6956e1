#!/bin/bash
# Round 6 evidence: the default bench command under rocprofv3 (kernel stats, then one --pmc pass per counter group - never together with
# a trace domain other than --kernel-trace), in the product's default arithmetic and in the opt-in selective one; the prefill PMC.
# Everything lands under gpurun_out/r6prof/ (copy the summaries into profiles/r06/).
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r6prof; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for A in reference selective; do
  B="python $R/bench.py --no-cpu-baseline --no-extras --regions 1 --arithmetic $A"
  D=$OUT/$A; mkdir -p $D
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats -o bench -- $B --steps 20 --warmup 5 > $D/bench_under_stats.txt 2>/dev/null
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $D/pmc_fetch -o bench -- $B --steps 5 --warmup 2 > /dev/null 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $D/pmc_write -o bench -- $B --steps 5 --warmup 2 > /dev/null 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $D/pmc_sq -o bench -- $B --steps 5 --warmup 2 > /dev/null 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $D/pmc_sq2 -o bench -- $B --steps 5 --warmup 2 > /dev/null 2>&1
  python $R/tools/pmc_summary.py $D $D/pmc_summary.json
  cp $D/stats/bench_kernel_stats.csv $OUT/bench_h8192_chain_${A}_kernel_stats.csv
  cut -c1-200 $D/stats/bench_kernel_stats.csv | head -4
  rm -rf $D/stats $D/pmc_fetch $D/pmc_write $D/pmc_sq $D/pmc_sq2     # (traces / databases: tens of MiB; gpurun merges back 64 MiB at most)
done
cp $OUT/reference/pmc_summary.json $OUT/bench_h8192_chain_exact_pmc_summary.json
cp $OUT/selective/pmc_summary.json $OUT/bench_h8192_chain_selective_pmc_summary.json
# the prefill route (BASELINE configs[3]: 8192 bf16 tokens through 4096x4096 and 14336x4096): kernel stats + MFMA-pipe busy
P=$OUT/prefill; mkdir -p $P
PB="python $R/tools/prefill_bench.py --tokens 8192 --shapes 4096,4096;4096,14336 --dtypes bf16"
timeout 600 $PB --out $P/prefill_bf16.json 2>&1 | grep -v amdgpu.ids > $P/prefill_bf16.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats -o pf -- $PB > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA --output-format csv -d $P/pmc -o pf -- $PB > /dev/null 2>&1
python $R/tools/pmc_kernels.py $P/pmc $OUT/prefill_bf16_pmc_kernels.json | cut -c1-300
cp $P/stats/pf_kernel_stats.csv $OUT/prefill_bf16_kernel_stats.csv
cp $P/prefill_bf16.txt $OUT/prefill_bf16_timings.txt
cp $P/prefill_bf16.json $OUT/prefill_bf16.json
cut -c1-160 $OUT/prefill_bf16_kernel_stats.csv | head -8
rm -rf $P
du -sh $R/gpurun_out
