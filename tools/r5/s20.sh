#!/bin/bash
# round 5, session 20: the round's final build - GPU suite, smoke, sliced fuzz in both dtypes (exact tokens, column parts), the bench
# line in the default arithmetic (+ --folded), rocprofv3 kernel stats of the same command
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r5s20; mkdir -p $OUT; rm -rf $OUT/stats
cd $R
export PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider --tb=short 2>&1 | tail -40 > $OUT/gpu_suite.txt
grep -E "^FAILED|passed|failed" $OUT/gpu_suite.txt | tail -12
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | tee $OUT/smoke.txt
timeout 600 python tools/gpu_fuzz.py --sliced --cases 50 --seed 91 2>&1 | grep -v amdgpu.ids > $OUT/fuzz_sliced_f16.txt; tail -1 $OUT/fuzz_sliced_f16.txt
timeout 400 python tools/gpu_fuzz.py --sliced --cases 25 --seed 92 --dtype bf16 2>&1 | grep -v amdgpu.ids > $OUT/fuzz_sliced_bf16.txt; tail -1 $OUT/fuzz_sliced_bf16.txt
timeout 700 python bench.py --steps 20 --warmup 5 > $OUT/bench_h8192_chain.json 2> $OUT/bench.err
python - <<'PY'
import json, os
try:
    d = json.load(open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r5s20/bench_h8192_chain.json")))
    print("bench", round(d["value"], 1), round(d["roofline"]["frac"], 4), d["config"]["kernel"], "traffic", d["roofline"].get("traffic"))
    for k, v in d["roofline"]["module_path"].items():
        print("  module_path", k, {kk: (round(vv, 3) if isinstance(vv, float) else vv) for kk, vv in v.items()})
    for k, v in d["extras"].items():
        if isinstance(v, dict):
            print(" ", k, {kk: (round(vv, 3) if isinstance(vv, float) else vv) for kk, vv in v.items() if kk in ("us_per_launch", "us_per_layer", "error", "GBps", "tokens_per_s", "vqlinear_us_per_token")},
                  {kk: round(v[kk]["us_per_layer"], 2) for kk in ("default", "sliced_layout", "exact_sliced_layout") if kk in v})
except Exception as e:
    print("bench failed", e); print(open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r5s20/bench.err")).read()[-2000:])
PY
timeout 300 python bench.py --folded --steps 20 --warmup 5 --no-extras > $OUT/bench_h8192_chain_folded.json 2>> $OUT/bench.err
python -c "
import json; d=json.load(open('$OUT/bench_h8192_chain_folded.json')); print('bench --folded', round(d['value'],1), round(d['roofline']['frac'],4))"
B="python $R/bench.py --no-cpu-baseline --no-extras --regions 1"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- $B --steps 20 --warmup 5 > /dev/null 2>&1
cd $R
find $OUT -name "bench_kernel_trace.csv" -delete; find $OUT -name "bench_agent_info.csv" -delete
f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1); cut -c1-200 $f | head -3
