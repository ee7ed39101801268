#!/bin/bash
# Round 4, GPU session 4: GPU suite after the warp-kernel consolidation (training on the quad kernels), bench lines, HBM traffic of GetCost
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/s4
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout 400 python bench.py --no-batch-sweep --no-cpu-baseline --steps 10 --warmup 2 --conv-table > $O/bench.json 2> $O/bench.err
timeout 600 python bench.py --config cfg4 --steps 5 --warmup 2 > $O/bench_cfg4.json 2> $O/bench_cfg4.err
timeout 600 python bench.py --config cfg5 --steps 5 --warmup 2 > $O/bench_cfg5.json 2> $O/bench_cfg5.err
cd /tmp
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-batch-sweep > $O/pmc_fetch.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-batch-sweep > $O/pmc_write.log 2>&1
cd $R
python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write r4 96 > $O/pmc_traffic.log 2>&1
cp profiles/r4_pmc_hbm_traffic_per_kernel.csv profiles/r4_getcost_traffic.json $O/ 2>/dev/null
rm -rf $O/pmc_fetch $O/pmc_write
echo done > $O/finished
