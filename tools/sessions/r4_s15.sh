#!/bin/bash
# Round 4, GPU session 15 (the last 6 GPU-minutes): the fused stem with conv0.0 on row pairs and its persistent grid from the occupancy
# query -- stem tests, end-to-end parity, smoke, one short bench run with the conv table
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/s15
mkdir -p $O
cd $R
timeout 240 python -m pytest tests/test_ops.py tests/test_model_gpu.py -q -m gpu -k "stem or golden_end_to_end or fresh_inputs or sizes_not_multiple or native_library or (full_size_against_oracle and cfg2)" > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
echo "smoke rc=$?" >> $O/smoke.log
timeout 200 python bench.py --no-batch-sweep --no-cpu-baseline --steps 10 --warmup 2 --conv-table > $O/bench.json 2> $O/bench.err
timeout 60 python tools/diag_r4.py stem > $O/stem.jsonl 2> $O/stem.err
echo done > $O/finished
