#!/bin/bash
# Round 4, GPU session 16: the step with the stem's 4-byte halo pieces (faster than the 16-byte form on random data since session 15's changes)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/s16
mkdir -p $O
cd $R
DMVS_STEM_V16=0 timeout 150 python bench.py --no-batch-sweep --no-cpu-baseline --steps 10 --warmup 2 > $O/bench_stem4.json 2> $O/bench_stem4.err
echo done > $O/finished
