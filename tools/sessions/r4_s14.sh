#!/bin/bash
# Round 4, GPU session 14: the closing measurements again, after the tall conv tiles and the training-graph caches -- GPU suite, smoke, the full default bench line, the profiled bench run, cfg3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/s14
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
echo "smoke rc=$?" >> $O/smoke.log
timeout 900 python bench.py > $O/bench_full.json 2> $O/bench_full.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $R/bench.py --no-batch-sweep --no-cpu-baseline --steps 6 --warmup 2 --conv-table > $O/bench_profiled.json 2> $O/bench_profiled.err
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/bench_b96_kernel_stats.csv 2>/dev/null
rm -rf $O/prof
cd $R
timeout 400 python bench.py --config cfg3 --steps 10 --warmup 2 > $O/bench_cfg3.json 2> $O/bench_cfg3.err
timeout 400 python bench.py --config cfg4 --steps 5 --warmup 2 > $O/bench_cfg4.json 2> $O/bench_cfg4.err
echo done > $O/finished
