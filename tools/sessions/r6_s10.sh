#!/bin/bash
# Round 6, session 10: split-bf16 (fp32-accurate) matrix arithmetic in the 2-D convolutions: GPU parity tests, then the step and the per-layer
# table with DMVS_CONV_ARITH=split (16 x 16 and forced 16 x 8 tiles) against the exact-fp32 kernels.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6_s10
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_ops.py -x -q -m gpu -k "split_bf16 or bf16_matrix" > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu.log
run() { name=$1; shift; env "$@" timeout 300 python bench.py --no-batch-sweep --no-cpu-baseline --no-probe --steps 8 --warmup 2 --conv-table > $O/bench_$name.json 2> $O/bench_$name.err; }
run fp32 DMVS_X=0
run split DMVS_CONV_ARITH=split
run split_mt2 DMVS_CONV_ARITH=split DMVS_CONV_MT=2
run fp32_b DMVS_X=0
echo done > $O/finished
