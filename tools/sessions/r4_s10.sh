#!/bin/bash
# Round 4, GPU session 10: batch 1 with / without the small-grid n-tile rule
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/s10
mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_ops.py -q -m gpu -k "small_grid or 1x1" > $O/pytest_gpu.log 2>&1
for B in 1 4; do
  timeout 300 python bench.py --batch $B --steps 30 --warmup 5 --no-batch-sweep --no-cpu-baseline > $O/bench_b${B}_rule.json 2> $O/bench_b${B}_rule.err
  DMVS_CONV_NT=max timeout 300 python bench.py --batch $B --steps 30 --warmup 5 --no-batch-sweep --no-cpu-baseline > $O/bench_b${B}_ntmax.json 2> $O/bench_b${B}_ntmax.err
done
echo done > $O/finished
