#!/bin/bash
# Round 4, GPU session 2: the GPU parity suite on the ABI-2 library, the pipelined GetCost against the round-3 loop, a short bench.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/s2
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout 400 python tools/diag_r4.py getcost > $O/getcost_ab_b96.jsonl 2> $O/getcost_ab.err
DIAG_B=16 timeout 300 python tools/diag_r4.py getcost > $O/getcost_ab_b16.jsonl 2>> $O/getcost_ab.err
timeout 600 python bench.py --no-batch-sweep --no-cpu-baseline --steps 10 --warmup 2 --conv-table > $O/bench_short.json 2> $O/bench_short.err
echo done > $O/finished
