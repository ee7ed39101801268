#!/bin/bash
# Round 4, GPU session 8: the 16-byte 1x1 form: parity on the GPU, bench with / without it
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/s8
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_ops.py tests/test_model_gpu.py -q -m gpu -k "conv2d or golden_end_to_end or a15 or a8" > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout 400 python bench.py --no-batch-sweep --no-cpu-baseline --steps 10 --warmup 2 --conv-table > $O/bench_px4.json 2> $O/bench_px4.err
DMVS_CONV1X1_PX4=0 timeout 400 python bench.py --no-batch-sweep --no-cpu-baseline --steps 10 --warmup 2 --conv-table > $O/bench_nopx4.json 2> $O/bench_nopx4.err
echo done > $O/finished
