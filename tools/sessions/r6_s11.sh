#!/bin/bash
# Round 6, session 11: split-bf16 (fp32-accurate) matrix arithmetic in the 2-D convolutions: the step and the per-layer table with
# --conv-arith split (16 x 16 and forced 16 x 8 tiles) against the exact-fp32 kernels.  (Session 10 set DMVS_CONV_ARITH, which bench.py's
# explicit args.conv_arith overrides: its three lines are all fp32.)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6_s11
mkdir -p $O
cd $R
run() { name=$1; shift; env "$@" > $O/bench_$name.json 2> $O/bench_$name.err; }
run split DMVS_X=0 timeout 300 python bench.py --no-batch-sweep --no-cpu-baseline --no-probe --steps 8 --warmup 2 --conv-table --conv-arith split
run split_mt2 DMVS_CONV_MT=2 timeout 300 python bench.py --no-batch-sweep --no-cpu-baseline --no-probe --steps 8 --warmup 2 --conv-table --conv-arith split
run fp32 DMVS_X=0 timeout 300 python bench.py --no-batch-sweep --no-cpu-baseline --no-probe --steps 8 --warmup 2 --conv-table
run bf16 DMVS_X=0 timeout 300 python bench.py --no-batch-sweep --no-cpu-baseline --no-probe --steps 8 --warmup 2 --conv-table --conv-arith bf16
echo done > $O/finished
