#!/bin/bash
# Round 4, GPU session 7: batch sweep of the headline configuration beyond 96 reference views per step
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/s7
mkdir -p $O
cd $R
for B in 96 128 160 192; do
  timeout 500 python bench.py --batch $B --no-batch-sweep --no-cpu-baseline --steps 8 --warmup 2 > $O/bench_b$B.json 2> $O/bench_b$B.err
done
echo done > $O/finished
