#!/bin/bash
# Round 4, GPU session 6: the quad-per-pixel backward of the warp kernels: parity on the GPU, training step against round 1's backward
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/s6
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_ops.py tests/test_train.py tests/test_trainer.py -q -m gpu -k "backward or train or many_source or step" > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout 600 python bench.py --config cfg4 --steps 5 --warmup 2 > $O/bench_cfg4_quad.json 2> $O/bench_cfg4_quad.err
DMVS_WARP_BWD=legacy timeout 600 python bench.py --config cfg4 --steps 5 --warmup 2 > $O/bench_cfg4_legacy.json 2> $O/bench_cfg4_legacy.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg4 -- python $R/bench.py --config cfg4 --steps 3 --warmup 1 > $O/prof_cfg4.log 2>&1
cp $(find $O/prof_cfg4 -name "*kernel_stats.csv" | head -1) $O/cfg4_kernel_stats.csv 2>/dev/null
rm -rf $O/prof_cfg4
echo done > $O/finished
