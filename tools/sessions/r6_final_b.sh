#!/bin/bash
# Round 6, after the last session: the same HEAD kernels (only profiles/ and docs changed since), the default bench line once more now that the PMC
# traffic file of THIS kernel source is committed (bench.py quotes it only when its sha matches), smoke()
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6_final_b
mkdir -p $O
cd $R
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
echo "smoke rc=$?" >> $O/smoke.log
timeout 500 python bench.py --conv-table > $O/bench_full.json 2> $O/bench_full.err
timeout 300 python bench.py --no-batch-sweep --no-cpu-baseline --no-probe --steps 10 --warmup 2 --conv-arith fp32 > $O/bench_fp32.json 2> $O/bench_fp32.err
echo done > $O/finished
