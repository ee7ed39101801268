#!/bin/bash
# Round 6, session 6: which tiles share an XCD's L2 in the tiled conv2d kernels (DMVS_TUNE_XCD_GROUP): the step and the per-layer table for
# plain round robin (1) against groups of 2 / 4 / 8 x-adjacent tiles, one / two tile rows, one image.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6_s6
mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_ops.py -x -q -m gpu -k "xcd_grouped" > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu.log
for c in 1 2 3 5 1 4 6 7; do
  DMVS_CONV_XCD=$c timeout 300 python bench.py --no-batch-sweep --no-cpu-baseline --no-probe --steps 8 --warmup 2 --conv-table > $O/bench_xcd${c}_$(date +%s).json 2> $O/bench_xcd${c}_$(date +%s).err
done
echo done > $O/finished
