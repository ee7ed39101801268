#!/bin/bash
# Round 5 (python-only change since the final sessions: bench.py --scene-mode now also takes --config cfg3): scene mode on BASELINE.json
# configs[2] -- one 49-view scene of 1152x864 images per step, bf16 feature storage and conv arithmetic
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5_scene_cfg3
mkdir -p $O
cd $R
timeout 300 python bench.py --scene-mode --config cfg3 --steps 6 --warmup 2 > $O/bench_scene_cfg3.json 2> $O/bench_scene_cfg3.err
echo done > $O/finished
