#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r3fuzz; mkdir -p $OUT
cd $R
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke.txt
timeout 900 python tools/gpu_fuzz.py --chains --cases 40 --seed 1 2>&1 | grep -v amdgpu.ids | tail -45 | tee $OUT/fuzz_chains_f16.txt
timeout 900 python tools/gpu_fuzz.py --chains --cases 20 --seed 2 --dtype bf16 2>&1 | grep -v amdgpu.ids | tail -24 | tee $OUT/fuzz_chains_bf16.txt
timeout 200 python bench.py --no-cpu-baseline --no-extras --steps 50 --regions 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['sclk_during_timed_regions'])" | tee $OUT/sclk.txt
