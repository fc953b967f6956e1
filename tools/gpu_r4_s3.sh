#!/bin/bash
# round 4, GPU call 3: full GPU suite, counting fuzz with the distinct-rows gate, the bench line (conditioning + several steps
# per graph) + rocprofv3 evidence for it (kernel stats, PMC passes) under gpurun_out/r4bench (copied into profiles/r04)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4bench; mkdir -p $OUT
cd $R
timeout 400 python -m pytest tests -q -m gpu -x 2>&1 | grep -v amdgpu.ids | tail -6 | tee $OUT/gpu_suite.txt
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke.txt
timeout 200 python tools/gpu_fuzz_count.py --layers 2048 --dtype f16 --chain 32 2>&1 | grep -v amdgpu.ids | tee $OUT/fuzz_count_f16_gated.txt
timeout 200 python tools/gpu_fuzz_count.py --layers 2048 --dtype bf16 --chain 32 --seed 1 2>&1 | grep -v amdgpu.ids | tee $OUT/fuzz_count_bf16_gated.txt
timeout 900 python bench.py 2> $OUT/bench_stderr.txt | tail -1 > $OUT/bench_h8192_chain.json
cut -c1-1200 $OUT/bench_h8192_chain.json; tail -3 $OUT/bench_stderr.txt
timeout 100 python bench.py --steps 20 --warmup 5 --no-extras 2> /dev/null | tail -1 > $OUT/bench_h8192_chain_driver_args.json
cut -c1-300 $OUT/bench_h8192_chain_driver_args.json
B="python $R/bench.py --no-cpu-baseline --no-extras --regions 1"
BS="python $R/bench.py --no-cpu-baseline --no-extras"   # the stats pass: the bench's own steps / regions
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- $BS > $OUT/bench_under_rocprofv3.json 2> /dev/null
export VPTQ_BENCH_CONDITIONING_S=0 VPTQ_BENCH_SOAK=0
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o bench -- $B --steps 5 --warmup 2 > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o bench -- $B --steps 5 --warmup 2 > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_sq -o bench -- $B --steps 5 --warmup 2 > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq2 -o bench -- $B --steps 5 --warmup 2 > /dev/null 2>&1
cd $R
python - <<'PY'
import csv, json, os, statistics
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r4bench")
rows = [r for r in csv.DictReader(open(os.path.join(out, "stats", "bench_kernel_trace.csv"))) if "gemv_k256c_kernel" in r["Kernel_Name"]]
d = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows)
json.dump({"kernel": "gemv_k256c_kernel (one dispatch = 32 layers)", "dispatches": len(d), "mean_us": statistics.mean(d),
           "median_us": statistics.median(d), "min_us": d[0], "max_us": d[-1], "p10_us": d[len(d) // 10], "p90_us": d[(9 * len(d)) // 10],
           "source": "rocprofv3 --kernel-trace: End_Timestamp - Start_Timestamp per dispatch of the stats run (conditioning, warm-up, timed regions and soak)"},
          open(os.path.join(out, "bench_h8192_chain_kernel_durations.json"), "w"), indent=1)
print(open(os.path.join(out, "bench_h8192_chain_kernel_durations.json")).read())
PY
rm -f $OUT/*/bench_kernel_trace.csv $OUT/*/bench_agent_info.csv
python tools/pmc_summary.py $OUT $OUT/bench_h8192_chain_pmc_summary.json
cut -c1-200 $OUT/stats/bench_kernel_stats.csv | head -4
