#!/bin/bash
# Round 6, session 12: split-bf16 arithmetic, second form (pre-split weights from global memory to registers, one fp32 staging buffer, no VALU in
# the matrix loop): GPU parity tests, the step and per-layer table against exact fp32.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6_s12
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_ops.py -x -q -m gpu -k "split_bf16 or bf16_matrix or conv2d_tall or conv2d_xcd" > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu.log
run() { name=$1; shift; env "$@" > $O/bench_$name.json 2> $O/bench_$name.err; }
run split DMVS_X=0 timeout 300 python bench.py --no-batch-sweep --no-cpu-baseline --no-probe --steps 8 --warmup 2 --conv-table --conv-arith split
run split_mt2 DMVS_CONV_MT=2 timeout 300 python bench.py --no-batch-sweep --no-cpu-baseline --no-probe --steps 8 --warmup 2 --conv-table --conv-arith split
run fp32 DMVS_X=0 timeout 300 python bench.py --no-batch-sweep --no-cpu-baseline --no-probe --steps 8 --warmup 2 --conv-table
echo done > $O/finished
