#!/bin/bash
# The N > 1 code path of bench.py on a ONE-GPU box: 2 (and 4) ranks share GPU 0, gloo collectives
# (VPTQ_BENCH_SAME_GPU=1).  Checks that the driver's multi-GPU command line runs end to end and prints
# one JSON line; the numbers mean nothing.
OUT=gpurun_out/r3ranks; mkdir -p $OUT
export VPTQ_BENCH_SAME_GPU=1
for N in 2 4; do
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29510 + N)) \
    bench.py --gpus $N --steps 3 --warmup 1 --regions 2 --tp-layers 2 > $OUT/bench_n${N}.json 2> $OUT/bench_n${N}.err
  echo "N=$N rc=$?"; tail -c 1500 $OUT/bench_n${N}.json; tail -5 $OUT/bench_n${N}.err
done
