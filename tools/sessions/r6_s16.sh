#!/bin/bash
# Round 6, session 16: split-bf16 arithmetic, fifth form (one accumulator: 124 registers, 4 waves per SIMD) + the dispatcher's rule: parity tests with
# the measured errors against fp64, the step with the rule and with the form forced everywhere.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6_s16
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_ops.py -x -q -m gpu -k "split_bf16" -s > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu.log
run() { name=$1; shift; env "$@" > $O/bench_$name.json 2> $O/bench_$name.err; }
run split_rule DMVS_X=0 timeout 300 python bench.py --no-batch-sweep --no-cpu-baseline --no-probe --steps 8 --warmup 2 --conv-table --conv-arith split
run split_all DMVS_CONV_SPLIT_ALL=1 timeout 300 python bench.py --no-batch-sweep --no-cpu-baseline --no-probe --steps 8 --warmup 2 --conv-table --conv-arith split
timeout 300 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "split_bf16" -s > $O/pytest_model.log 2>&1
echo "pytest rc=$?" >> $O/pytest_model.log
echo done > $O/finished
