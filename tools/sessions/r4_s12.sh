#!/bin/bash
# Round 4, GPU session 12: three convolution experiments per layer (tall tiles, 8-byte stride-2 operand reads, 5x5 row unrolling), the
# step with tall tiles against a same-box control, the training step with the per-step weight / embedding caches, the new GPU tests
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/s12
mkdir -p $O
cd $R
timeout 300 python tools/diag_r4.py convexp > $O/convexp.jsonl 2> $O/convexp.err
DMVS_CONV_TALL=1 timeout 300 python bench.py --no-batch-sweep --no-cpu-baseline --steps 10 --warmup 2 --conv-table > $O/bench_tall.json 2> $O/bench_tall.err
timeout 300 python bench.py --no-batch-sweep --no-cpu-baseline --steps 10 --warmup 2 --conv-table > $O/bench_control.json 2> $O/bench_control.err
timeout 400 python bench.py --config cfg4 --steps 5 --warmup 2 > $O/bench_cfg4.json 2> $O/bench_cfg4.err
timeout 600 python -m pytest tests/test_ops.py tests/test_train.py -q -m gpu -k "tall_tiles or gate_product or training_step" > $O/pytest_new.log 2>&1
echo "pytest rc=$?" >> $O/pytest_new.log
echo done > $O/finished
