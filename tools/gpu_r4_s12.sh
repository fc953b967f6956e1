#!/bin/bash
# round 4, GPU call 12: rows per wave (= workgroups per launch) of the sliced kernel, same box; 2 ranks on one GPU (bench --gpus 2 path)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4s12; mkdir -p $OUT
cd $R
S="8192,8192;4096,14336"
for kr in 0 256 65536; do
  for rpw in 0 1 2 4 8; do
    timeout 200 python tools/sliced_bench.py --kr $kr --rpw $rpw --shapes "$S" 2>&1 | grep -v amdgpu.ids | cut -c1-400 | sed "s/^/rpw$rpw /" | tee -a $OUT/sliced_rows_per_wave.txt
  done
done
timeout 300 bash tools/gpu_bench_2ranks.sh 2>&1 | tail -5 | cut -c1-400 | tee $OUT/bench_2ranks.txt
