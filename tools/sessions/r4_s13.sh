#!/bin/bash
# Round 4, GPU session 13: the tall-tile form with two n-tiles per layer, and the step with it forced everywhere against the new default
# (tall tiles for the one-n-tile layers only), same box
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/s13
mkdir -p $O
cd $R
timeout 300 python tools/diag_r4.py convexp2 > $O/convexp2.jsonl 2> $O/convexp2.err
DMVS_CONV_TALL=1 timeout 300 python bench.py --no-batch-sweep --no-cpu-baseline --steps 10 --warmup 2 --conv-table > $O/bench_tall2.json 2> $O/bench_tall2.err
timeout 300 python bench.py --no-batch-sweep --no-cpu-baseline --steps 10 --warmup 2 --conv-table > $O/bench_default.json 2> $O/bench_default.err
echo done > $O/finished
