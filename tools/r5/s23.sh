#!/bin/bash
# round 5, session 23: the bench line of the last build with the two extras added last (2 / 3 exact tokens in one pass; the 28672-column
# layer in column parts)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r5s23; mkdir -p $OUT
cd $R
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_h8192_chain.json 2> $OUT/bench.err
python - <<'PY'
import json, os
d = json.load(open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r5s23/bench_h8192_chain.json")))
print("bench", round(d["value"], 1), round(d["roofline"]["frac"], 4), d["config"]["kernel"])
for k in ("k65536_r256", "k65536_r65536", "v16_k65536_r65536", "k65536_r256_28672x8192"):
    v = d["extras"].get(k, {})
    print(k, {kk: (vv if not isinstance(vv, dict) else {a: (round(b, 2) if isinstance(b, float) else b) for a, b in vv.items() if a != "what"}) for kk, vv in v.items() if kk.startswith(("exact_tokens", "error", "default", "exact_sliced", "column_parts", "bit_identical"))})
PY
tail -3 $OUT/bench.err
