#!/bin/bash
# round 5, session 1: the one-hop cross-slice hand-over of gemv_sliced (fixed-point accumulator words) against the round-4
# epilogue on the same box; phase trace; the measured folded-form gate; full GPU suite; bench baseline with the new fields
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r5s1; mkdir -p $OUT
cd $R
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider --tb=short 2>&1 | tail -25 > $OUT/suite.txt
tail -3 $OUT/suite.txt
for kr in 0 256 65536; do
  for lib in default epi0; do
    L=""; [ $lib != default ] && L=$R/tools/_build/libvptq_hip_$lib.so
    echo "== kr=$kr lib=$lib" >> $OUT/sliced_ab.txt
    VPTQ_HIP_LIB=$L timeout 200 python tools/sliced_bench.py --kr $kr --shapes "8192,8192;4096,4096;4096,14336;14336,4096;8192,28672" 2>&1 | grep -v amdgpu.ids >> $OUT/sliced_ab.txt
  done
done
for lib in default epi0; do
  L=""; [ $lib != default ] && L=$R/tools/_build/libvptq_hip_$lib.so
  echo "== v16 kr=65536 lib=$lib" >> $OUT/sliced_ab.txt
  VPTQ_HIP_LIB=$L timeout 200 python tools/sliced_bench.py --v 16 --kr 65536 --shapes "8192,8192;4096,4096" 2>&1 | grep -v amdgpu.ids >> $OUT/sliced_ab.txt
done
cat $OUT/sliced_ab.txt | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('=='): print(l.strip()); continue
    try: r = json.loads(l)
    except Exception: continue
    print(f\"  {r['I']}x{r['O']} default {r['default_us']:.2f} sliced {r['sliced_us']:.2f} rel {r['rel_diff']:.1e}\")
"
for kr in 0 256; do
  VPTQ_HIP_LIB=$R/tools/_build/libvptq_hip_tr1.so timeout 120 python tools/sliced_trace.py --kr $kr 2>&1 | grep -v amdgpu.ids > $OUT/trace_kr$kr.json
done
python - <<'PY'
import json, os
for kr in (0, 256):
    try:
        d = json.load(open(os.path.join(os.environ["GRAFT_REPO_ROOT"], f"gpurun_out/r5s1/trace_kr{kr}.json")))
        print("trace kr", kr, d["runs"][-1])
    except Exception as e:
        print("trace kr", kr, "failed", e)
PY
timeout 400 python tools/gpu_gate_count.py --layers 1000 --dtype f16 --max-elems 20e6 2>&1 | grep -v amdgpu.ids | tee $OUT/gate_count_f16.txt | tail -14
timeout 300 python tools/gpu_gate_count.py --layers 500 --dtype bf16 --max-elems 20e6 2>&1 | grep -v amdgpu.ids | tee $OUT/gate_count_bf16.txt | tail -14
timeout 400 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
python - <<'PY'
import json, os
try:
    d = json.load(open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r5s1/bench.json")))
    print("bench", d["value"], d["roofline"]["frac"], json.dumps(d["roofline"].get("module_path")))
    for k, v in d["extras"].items():
        if isinstance(v, dict):
            print(" ", k, {kk: (round(vv, 3) if isinstance(vv, float) else vv) for kk, vv in v.items() if kk in ("us_per_launch", "us_per_layer", "error", "GBps")},
                  {kk: round(v[kk]["us_per_layer"], 2) for kk in ("default", "sliced_layout") if kk in v})
except Exception as e:
    print("bench failed", e)
PY
