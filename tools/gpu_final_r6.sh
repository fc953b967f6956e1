R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
# 1. the default bench command (what the driver runs), then the same with every extra
timeout 600 python bench.py > $O/bench_line_final.json 2> $O/bench_stderr.txt; cp $O/bench_full.json $O/bench_full_final.json 2>/dev/null
head -c 600 $O/bench_line_final.json; echo
timeout 1500 python bench.py --extras > $O/bench_line_final_extras.json 2>/dev/null; cp $O/bench_full.json $O/bench_full_final_extras.json 2>/dev/null
# 2. rocprofv3: kernel stats + PMC of the same command, default and selective
bash tools/gpu_profile_r6.sh > $O/gpu_profile_r6.log 2>&1; tail -5 $O/gpu_profile_r6.log
# 3. model level
for A in reference selective; do for D in f16 bf16; do
  VPTQ_ARITHMETIC=$A timeout 900 python tools/llama_decode.py --fuse --dtype $D 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$A $D', d['model'], round(d['decode_tok_s_hipgraph'],1), 'tok/s', round(d['vqlinear_us_per_token'],1), 'us VQuantLinear', round(d['ttft_ms'],1), 'ms TTFT')"
done; done > $O/llama_decode_final.txt 2>&1
cat $O/llama_decode_final.txt
# 4. the GPU suite
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -4 > $O/gpu_suite.txt; cat $O/gpu_suite.txt
du -sh $O
