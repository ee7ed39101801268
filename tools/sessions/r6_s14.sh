#!/bin/bash
# Round 6, session 14: split-bf16 arithmetic, fourth form (weights prefetched into alternating register sets without copies, at most two n-tiles per workgroup): parity tests, the step and per-layer table.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6_s14
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_ops.py -x -q -m gpu -k "split_bf16" > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu.log
run() { name=$1; shift; env "$@" > $O/bench_$name.json 2> $O/bench_$name.err; }
run split DMVS_X=0 timeout 300 python bench.py --no-batch-sweep --no-cpu-baseline --no-probe --steps 8 --warmup 2 --conv-table --conv-arith split
echo done > $O/finished
