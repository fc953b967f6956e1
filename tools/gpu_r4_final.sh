#!/bin/bash
# round 4, last bench of the round on the final build: the full bench line (with extras), the driver's arguments, rocprofv3
# kernel stats of the same command
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4final; mkdir -p $OUT
cd $R
( time timeout 900 python bench.py 2> $OUT/bench_stderr.txt | tail -1 > $OUT/bench_h8192_chain.json ) 2>&1 | grep real | tee $OUT/bench_wall.txt
cut -c1-900 $OUT/bench_h8192_chain.json; tail -3 $OUT/bench_stderr.txt
( time timeout 300 python bench.py --steps 20 --warmup 5 2> /dev/null | tail -1 > $OUT/bench_h8192_chain_driver_args.json ) 2>&1 | grep real | tee -a $OUT/bench_wall.txt
cut -c1-300 $OUT/bench_h8192_chain_driver_args.json
BS="python $R/bench.py --no-cpu-baseline --no-extras"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- $BS > $OUT/bench_under_rocprofv3.json 2> /dev/null
cd $R
python - <<'PY'
import csv, json, os, statistics
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r4final")
rows = [r for r in csv.DictReader(open(os.path.join(out, "stats", "bench_kernel_trace.csv"))) if "gemv_k256c_kernel" in r["Kernel_Name"]]
d = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows)
json.dump({"kernel": "gemv_k256c_kernel (one dispatch = 32 layers)", "dispatches": len(d), "mean_us": statistics.mean(d),
           "median_us": statistics.median(d), "min_us": d[0], "max_us": d[-1], "p10_us": d[len(d) // 10], "p90_us": d[(9 * len(d)) // 10],
           "source": "rocprofv3 --kernel-trace: End_Timestamp - Start_Timestamp per dispatch of the stats run"},
          open(os.path.join(out, "bench_h8192_chain_kernel_durations.json"), "w"), indent=1)
print(open(os.path.join(out, "bench_h8192_chain_kernel_durations.json")).read())
PY
rm -f $OUT/*/bench_kernel_trace.csv $OUT/*/bench_agent_info.csv
cut -c1-200 $OUT/stats/bench_kernel_stats.csv | head -4
