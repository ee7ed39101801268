#!/bin/bash
# Round 6, session 2: (a) the start-up stagger of the tiled conv2d kernels (DMVS_TUNE_STAGGER) against the lockstep reading of the conv table's
# `sum` column, per layer shape; (b) the interleaved channel -> lane mapping of the warp backward's atomics (DMVS_TUNE_BWD_INTERLEAVED) on the
# cfg4 training step; (c) the GPU tests of what changed since session 1.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6_s2
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_ops.py tests/test_scene.py tests/test_eval_gpu.py -x -q -m gpu -k "backward or getcost or scene_features or epilogue or a5 or eval_defaults or mask_upsample" > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu.log
for n in 0 1 2 4; do
  DMVS_CONV_STAGGER=$n timeout 300 python bench.py --no-batch-sweep --no-cpu-baseline --no-probe --steps 10 --warmup 3 --conv-table > $O/bench_stagger$n.json 2> $O/bench_stagger$n.err
done
for il in 0 1; do
  DMVS_BWD_INTERLEAVED=$il timeout 400 python bench.py --config cfg4 --steps 8 --warmup 2 > $O/bench_cfg4_il$il.json 2> $O/bench_cfg4_il$il.err
done
echo done > $O/finished
