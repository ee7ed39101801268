#!/bin/bash
# Round 4, GPU session 3: the GPU parity suite (all tests, no early stop), the lean conv2d specialisation against the generic kernel
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/s3
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout 400 python bench.py --no-batch-sweep --no-cpu-baseline --steps 10 --warmup 2 --conv-table > $O/bench_lean.json 2> $O/bench_lean.err
DMVS_CONV_LEAN=0 timeout 400 python bench.py --no-batch-sweep --no-cpu-baseline --steps 10 --warmup 2 --conv-table > $O/bench_nolean.json 2> $O/bench_nolean.err
echo done > $O/finished
