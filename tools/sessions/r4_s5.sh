#!/bin/bash
# Round 4, GPU session 5: where batch 1 and the training step spend their time (rocprofv3 kernel stats), bench after the view-aggregate /
# lean-GN / plane-store changes, HBM traffic of GetCost
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/s5
mkdir -p $O
cd $R
timeout 400 python bench.py --no-batch-sweep --no-cpu-baseline --steps 10 --warmup 2 --conv-table > $O/bench.json 2> $O/bench.err
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b1 -- python $R/bench.py --batch 1 --steps 30 --warmup 5 --no-batch-sweep --no-cpu-baseline > $O/prof_b1.log 2>&1
cp $(find $O/prof_b1 -name "*kernel_stats.csv" | head -1) $O/b1_kernel_stats.csv 2>/dev/null
rm -rf $O/prof_b1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg4 -- python $R/bench.py --config cfg4 --steps 3 --warmup 1 > $O/prof_cfg4.log 2>&1
cp $(find $O/prof_cfg4 -name "*kernel_stats.csv" | head -1) $O/cfg4_kernel_stats.csv 2>/dev/null
rm -rf $O/prof_cfg4
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-batch-sweep > $O/pmc_fetch.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-batch-sweep > $O/pmc_write.log 2>&1
cd $R
python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write r4 96 > $O/pmc_traffic.log 2>&1
cp profiles/r4_pmc_hbm_traffic_per_kernel.csv profiles/r4_getcost_traffic.json $O/ 2>/dev/null
rm -rf $O/pmc_fetch $O/pmc_write
echo done > $O/finished
