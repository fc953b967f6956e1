#!/bin/bash
# round 5, session 7: software-pipelined gathers in gemv_sliced (both arithmetics) against the same build without;
# f3 A/B at 64 / 256 tokens (fused GEMM vs dequant + hipBLASLt vs the batched-decode kernels); Llama-3-8B-shaped decode in the
# format of most published checkpoints (v8-k65536-256) in the default arithmetic, exact sliced kernel vs gather kernel
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r5s7; mkdir -p $OUT
cd $R
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gemv_sliced_gpu.py -m gpu -q -p no:cacheprovider --tb=short 2>&1 | tail -15 > $OUT/sliced_tests.txt
tail -4 $OUT/sliced_tests.txt
ab() {  # label, env, extra args
  echo "== $1" >> $OUT/sliced_pipe_ab.txt
  env $2 timeout 300 python tools/sliced_bench.py $3 --shapes "8192,8192;4096,4096;4096,14336;14336,4096" 2>&1 | grep -v amdgpu.ids >> $OUT/sliced_pipe_ab.txt
}
for kr in 0 256; do
  ab "exact kr=$kr pipelined" "X=1" "--exact --kr $kr"
  ab "exact kr=$kr not pipelined" "VPTQ_HIP_LIB=$R/tools/_build/libvptq_hip_np.so" "--exact --kr $kr"
  ab "folded kr=$kr pipelined" "X=1" "--kr $kr"
  ab "folded kr=$kr not pipelined" "VPTQ_HIP_LIB=$R/tools/_build/libvptq_hip_np.so" "--kr $kr"
done
ab "folded kr=65536 pipelined" "X=1" "--kr 65536"
ab "folded kr=65536 not pipelined" "VPTQ_HIP_LIB=$R/tools/_build/libvptq_hip_np.so" "--kr 65536"
cat $OUT/sliced_pipe_ab.txt | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('=='): print(l.strip()); continue
    try: r = json.loads(l)
    except Exception: print(l.strip()[:200]); continue
    print(f\"  {r['I']}x{r['O']} gather {r['default_us']:.2f} sliced {r['sliced_us']:.2f} slices {r['slices']} rel {r['rel_diff']:.1e}\")
"
timeout 300 python tools/prefill_bench.py --tokens 17,32,64,128,256,512 --shapes "8192,8192;4096,4096" --dtypes f16,bf16 2>&1 | grep -v amdgpu.ids > $OUT/f3_fused_vs_dense_17_to_512_tokens.txt
timeout 200 python tools/tokens_bench.py --shapes "8192,8192;4096,4096" --tokens 16,32,48,64 2>&1 | grep -v amdgpu.ids > $OUT/f3_batched_decode_16_to_64_tokens.txt
VPTQ_ARITHMETIC=folded timeout 200 python tools/tokens_bench.py --shapes "8192,8192;4096,4096" --tokens 16,32,48,64 2>&1 | grep -v amdgpu.ids > $OUT/f3_batched_decode_16_to_64_tokens_folded.txt
python - <<'PY'
import json, os
R = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r5s7/"
for l in open(R + "f3_fused_vs_dense_17_to_512_tokens.txt"):
    try: r = json.loads(l)
    except Exception: continue
    print(f"f3 {r['dtype']} {r['I']}x{r['O']} T={r['tokens']}: fused {r['fused_us']:.1f} dense {r['dense_us']:.1f} (dequant {r['dequant_us']:.1f} + gemm {r['gemm_us']:.1f})")
for f in ("f3_batched_decode_16_to_64_tokens.txt", "f3_batched_decode_16_to_64_tokens_folded.txt"):
    for l in open(R + f):
        try: r = json.loads(l)
        except Exception: continue
        print(f, r["I"], r["tokens"], round(r["us"], 1), r["kernel"])
PY
for mode in auto 0; do
  VPTQ_SLICED_LAYOUT=$mode timeout 400 python tools/llama_decode.py --fuse --k 65536 --kr 256 --new 128 2>&1 | grep "^{" | tail -1 > $OUT/llama8b_k65536_r256_reference_sliced_$mode.json
  python -c "
import json; d=json.load(open('$OUT/llama8b_k65536_r256_reference_sliced_$mode.json')); print('llama k65536-256 reference arithmetic, VPTQ_SLICED_LAYOUT=$mode:', round(d['decode_tok_s_hipgraph'],1), 'tok/s, vqlinear us/token', round(d.get('vqlinear_us_per_token') or 0,1))"
done
