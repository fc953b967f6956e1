#!/bin/bash
# round 5, session 12: second queue stage of the two-table exact layouts decoupled from the first (8 blocks of element words, 2 of gathers)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r5s12; mkdir -p $OUT
cd $R
for a in "--exact --kr 65536" "--exact --v 16 --kr 65536" "--exact --kr 4096" "--exact --v 16 --kr 4096"; do
  echo "== $a" >> $OUT/rg.txt
  timeout 200 python tools/sliced_bench.py $a --shapes "8192,8192;4096,14336;14336,4096" --ring 8 2>&1 | tail -8 >> $OUT/rg.txt
done
cat $OUT/rg.txt
