#!/bin/bash
# Round 5, the LAST GPU session (second take: a comment in warp_quad.hip was corrected after r5_final.sh, same binary) on the commit that is
# HEAD -- no csrc/ or include/ change after it: the whole GPU suite with -x exactly as the driver runs it, smoke(), the full default bench line,
# the rocprofv3 kernel stats of the same command, the PMC traffic passes and the GetCost ceiling probe on this kernel source, cfg3 / cfg5 /
# scene-mode lines (cfg4 and cfg5 at batch 8 / 16: r5_final.sh, identical kernels)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5_final2
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/ -x -q -m gpu > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
echo "smoke rc=$?" >> $O/smoke.log
timeout 400 python bench.py --conv-table > $O/bench_full.json 2> $O/bench_full.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b96 -- python $R/bench.py --steps 5 --warmup 2 --no-batch-sweep --no-cpu-baseline > $O/prof_b96_line.json 2> $O/prof_b96.err
cp $(find $O/prof_b96 -name "*kernel_stats.csv" | head -1) $O/b96_kernel_stats.csv 2>/dev/null
rm -rf $O/prof_b96
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-batch-sweep > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-batch-sweep > $O/pmc_write.log 2>&1
cd $R
python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write r5 96 > $O/pmc_traffic.log 2>&1
cp profiles/r5_pmc_hbm_traffic_per_kernel.csv profiles/r5_getcost_traffic.json $O/ 2>/dev/null
rm -rf $O/pmc_fetch $O/pmc_write
timeout 300 python tools/diag_r4.py getcost > $O/getcost_probe.jsonl 2> $O/getcost_probe.err
timeout 200 python bench.py --config cfg3 --steps 10 --warmup 2 > $O/bench_cfg3.json 2> $O/bench_cfg3.err
timeout 200 python bench.py --config cfg5 --steps 6 --warmup 2 > $O/bench_cfg5_b2.json 2> $O/bench_cfg5_b2.err
timeout 200 python bench.py --scene-mode --steps 8 --warmup 2 > $O/bench_scene.json 2> $O/bench_scene.err
echo done > $O/finished
