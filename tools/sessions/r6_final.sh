#!/bin/bash
# Round 6, the LAST GPU session on the commit that is HEAD -- no csrc/ or include/ change after it: the whole GPU suite with -x exactly as the
# driver runs it, smoke(), the full default bench line, the rocprofv3 kernel stats of the same command, the PMC traffic passes on this kernel
# source, the batch-1 kernel stats, cfg3 / cfg4 / cfg5 / scene-mode lines.  tools/ingest_final.py copies the artefacts into profiles/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6_final
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/ -x -q -m gpu > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
echo "smoke rc=$?" >> $O/smoke.log
timeout 500 python bench.py --conv-table > $O/bench_full.json 2> $O/bench_full.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b96 -- python $R/bench.py --steps 5 --warmup 2 --no-batch-sweep --no-cpu-baseline > $O/prof_b96_line.json 2> $O/prof_b96.err
cp $(find $O/prof_b96 -name "*kernel_stats.csv" | head -1) $O/b96_kernel_stats.csv 2>/dev/null
rm -rf $O/prof_b96
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-batch-sweep --no-probe > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-batch-sweep --no-probe > $O/pmc_write.log 2>&1
cd $R
python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write r6 96 > $O/pmc_traffic.log 2>&1
cp profiles/r6_pmc_hbm_traffic_per_kernel.csv profiles/r6_getcost_traffic.json $O/ 2>/dev/null
rm -rf $O/pmc_fetch $O/pmc_write
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b1 -- python $R/bench.py --batch 1 --steps 30 --warmup 5 --no-batch-sweep --no-cpu-baseline --no-probe > $O/prof_b1_line.json 2> $O/prof_b1.err
cp $(find $O/prof_b1 -name "*kernel_stats.csv" | head -1) $O/b1_kernel_stats.csv 2>/dev/null
rm -rf $O/prof_b1
cd $R
timeout 300 python bench.py --config cfg4 --steps 10 --warmup 3 > $O/bench_cfg4.json 2> $O/bench_cfg4.err
timeout 200 python bench.py --config cfg3 --steps 10 --warmup 2 > $O/bench_cfg3.json 2> $O/bench_cfg3.err
timeout 200 python bench.py --config cfg5 --steps 6 --warmup 2 > $O/bench_cfg5_b2.json 2> $O/bench_cfg5_b2.err
timeout 200 python bench.py --scene-mode --steps 8 --warmup 2 > $O/bench_scene.json 2> $O/bench_scene.err
echo done > $O/finished
