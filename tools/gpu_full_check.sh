#!/bin/bash
# full GPU validation: parity suite (default and VPTQ_EXACT=1), smoke, the default bench line (stdout must be ONE line)
OUT=gpurun_out/r5e; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | grep -E "AssertionError|assert |passed|failed|Error" | head -20 | tee $OUT/tests.txt
VPTQ_TUNING=1 VPTQ_EXACT=1 timeout 1200 python -m pytest tests -m gpu -q 2>&1 | grep -E "AssertionError|assert |passed|failed|Error" | head -20 | tee $OUT/tests_exact.txt
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1 | tee $OUT/smoke.txt
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$? stdout lines: $(wc -l < $OUT/bench.json)"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5e/bench.json"))
print({k: d[k] for k in ("value", "ms_per_step")}, d["roofline"]["frac"], d["roofline"]["us_per_launch"])
print({k: (v.get("GBps"), v.get("us_per_launch")) for k, v in d["extras"].items() if isinstance(v, dict) and "GBps" in v})
print(d["extras"].get("llama3_8b_decode")); print(d["cpu_baseline"])
PY
