#!/bin/bash
# Round 6, session 1: the GPU suite on the refactored warp kernels (warp_quad_core.h) + the fused mask head / convex upsampling kernel, the full
# default bench line with the in-process ceiling probe and the random-line-gather calibration, and the mask-fusion A/B on the same box.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6_s1
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/ -x -q -m gpu > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout 500 python bench.py --conv-table > $O/bench_full.json 2> $O/bench_full.err
DMVS_MASK_FUSE=0 timeout 300 python bench.py --no-batch-sweep --no-cpu-baseline --no-probe --steps 10 --warmup 3 > $O/bench_nofuse.json 2> $O/bench_nofuse.err
timeout 300 python bench.py --no-batch-sweep --no-cpu-baseline --no-probe --steps 10 --warmup 3 > $O/bench_fuse.json 2> $O/bench_fuse.err
echo done > $O/finished
