#!/bin/bash
# Round 5, after the last session: the same HEAD kernels (only profiles/ and docs changed since), the default bench line once more now that
# the PMC traffic file and the ceiling probe of THIS kernel source are committed (bench.py quotes them only when their sha matches), smoke()
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5_final_b
mkdir -p $O
cd $R
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
echo "smoke rc=$?" >> $O/smoke.log
timeout 400 python bench.py --conv-table > $O/bench_full.json 2> $O/bench_full.err
echo done > $O/finished
