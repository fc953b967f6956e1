#!/bin/bash
# first GPU session: parity tests + kernel A/B
set -x
mkdir -p gpurun_out
python -c "import torch; print(torch.__version__, torch.cuda.get_device_name(0))" 2>&1 | tail -1
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 > gpurun_out/pytest_gpu.log
cat gpurun_out/pytest_gpu.log | tail -30
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3
for H in 8192 4096; do
  timeout 300 python tools/microbench.py --hidden $H --out gpurun_out/mb_${H}_default.json 2>&1 | tail -5
  VPTQ_K256_TAB=0 timeout 300 python tools/microbench.py --hidden $H --out gpurun_out/mb_${H}_tab0.json 2>&1 | tail -5
done
VPTQ_K256_REDUCE=0 timeout 300 python tools/microbench.py --hidden 8192 --out gpurun_out/mb_8192_red0.json 2>&1 | tail -5
