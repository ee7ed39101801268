#!/bin/bash
# Round 6, session 5: FeatureNet depth-first over chunks of reference views (do the inter-layer activations stay in the 256 MB Infinity Cache?)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6_s5
mkdir -p $O
cd $R
for c in 0 48 24 12 6 3; do
  DMVS_FEAT_CHUNK=$c timeout 300 python bench.py --no-batch-sweep --no-cpu-baseline --no-probe --steps 8 --warmup 2 > $O/bench_chunk$c.json 2> $O/bench_chunk$c.err
done
echo done > $O/finished
