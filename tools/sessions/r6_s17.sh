#!/bin/bash
# Round 6, session 17: split-bf16 arithmetic as the engine's default (rule: stride 1 with >= 16 input channels, stride 2 with >= 25 taps and >= 32
# input channels; channel-last fp32 outputs included): the step (default and --conv-arith fp32), then the WHOLE GPU suite on the new default.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6_s17
mkdir -p $O
cd $R
run() { name=$1; shift; env "$@" > $O/bench_$name.json 2> $O/bench_$name.err; }
run default DMVS_X=0 timeout 300 python bench.py --no-batch-sweep --no-cpu-baseline --no-probe --steps 8 --warmup 2 --conv-table
run fp32 DMVS_X=0 timeout 300 python bench.py --no-batch-sweep --no-cpu-baseline --no-probe --steps 8 --warmup 2 --conv-table --conv-arith fp32
timeout 1500 python -m pytest tests/ -x -q -m gpu > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu.log
echo done > $O/finished
