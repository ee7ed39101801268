#!/bin/bash
# Round 5, GPU session 5 (run on commit ecf0108; the feature was reverted afterwards -- profiles/r5_multistream_ab.json): the side-stream forward (Engine.multistream) -- bit-identity tests, then the batch sweep with the graph's parallel
# branches next to the single-stream graph; the cfg4 line with its new roofline blocks
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5_s5
mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_model_gpu.py tests/test_scene.py tests/test_eval_gpu.py -q -m gpu -k "side_stream or hip_graph or scene or eval" > $O/pytest_ms.log 2>&1
echo "pytest rc=$?" >> $O/pytest_ms.log
timeout 300 python bench.py --no-cpu-baseline --steps 6 --warmup 2 > $O/bench_sweep.json 2> $O/bench_sweep.err
DMVS_MULTISTREAM=1 timeout 200 python bench.py --no-cpu-baseline --no-batch-sweep --steps 10 --warmup 2 > $O/bench_ms_b96.json 2> $O/bench_ms_b96.err
timeout 200 python bench.py --config cfg4 --steps 5 --warmup 2 > $O/bench_cfg4.json 2> $O/bench_cfg4.err
echo done > $O/finished
