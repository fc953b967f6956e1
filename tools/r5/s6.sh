#!/bin/bash
# round 5, session 6: the round's evidence under the new default arithmetic (reference roundings): GPU suite, the bench line,
# rocprofv3 kernel stats + PMC passes of the same command, the counting fuzz of both arithmetics through the product route
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r5s6; mkdir -p $OUT
cd $R
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --tb=short 2>&1 | tail -40 > $OUT/gpu_suite.txt
grep -E "^FAILED|passed|failed" $OUT/gpu_suite.txt | tail -12
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $OUT/smoke.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_h8192_chain.json 2> $OUT/bench.err
python - <<'PY'
import json, os
try:
    d = json.load(open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r5s6/bench_h8192_chain.json")))
    print("bench", round(d["value"], 1), round(d["roofline"]["frac"], 4), d["config"]["kernel"], d["config"]["arithmetic"][:40])
    print(json.dumps(d["roofline"].get("module_path"), indent=0)[:3000])
    print("cpu_baseline", d.get("cpu_baseline"))
    for k, v in d["extras"].items():
        if isinstance(v, dict):
            print(" ", k, {kk: (round(vv, 3) if isinstance(vv, float) else vv) for kk, vv in v.items() if kk in ("us_per_launch", "us_per_layer", "error", "GBps", "tokens_per_s", "vqlinear_us_per_token")},
                  {kk: round(v[kk]["us_per_layer"], 2) for kk in ("default", "sliced_layout") if kk in v})
except Exception as e:
    print("bench failed", e); print(open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r5s6/bench.err")).read()[-2000:])
PY
B="python $R/bench.py --no-cpu-baseline --no-extras --regions 1"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- $B --steps 20 --warmup 5 > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o bench -- $B --steps 5 --warmup 2 > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o bench -- $B --steps 5 --warmup 2 > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_sq -o bench -- $B --steps 5 --warmup 2 > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq2 -o bench -- $B --steps 5 --warmup 2 > /dev/null 2>&1
cd $R
find $OUT -name "bench_kernel_trace.csv" -delete; find $OUT -name "bench_agent_info.csv" -delete
python tools/pmc_summary.py $OUT $OUT/bench_h8192_chain_exact_pmc_summary.json
find $OUT -name "*kernel_stats.csv" | head -2; f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1); cut -c1-200 $f | head -4
VPTQ_ARITHMETIC=folded timeout 500 python tools/gpu_gate_count.py --layers 4096 --dtype f16 --max-elems 20e6 2>&1 | grep -v amdgpu.ids > $OUT/gate_count_f16_folded_opt_in.txt; tail -14 $OUT/gate_count_f16_folded_opt_in.txt
VPTQ_ARITHMETIC=folded timeout 300 python tools/gpu_gate_count.py --layers 2048 --dtype bf16 --max-elems 20e6 2>&1 | grep -v amdgpu.ids > $OUT/gate_count_bf16_folded_opt_in.txt; tail -3 $OUT/gate_count_bf16_folded_opt_in.txt
timeout 500 python tools/gpu_gate_count.py --layers 4096 --dtype f16 --max-elems 20e6 2>&1 | grep -v amdgpu.ids > $OUT/gate_count_f16_reference_default.txt; tail -14 $OUT/gate_count_f16_reference_default.txt
timeout 500 python tools/gpu_gate_count.py --layers 4096 --dtype bf16 --max-elems 20e6 2>&1 | grep -v amdgpu.ids > $OUT/gate_count_bf16_reference_default.txt; tail -14 $OUT/gate_count_bf16_reference_default.txt
