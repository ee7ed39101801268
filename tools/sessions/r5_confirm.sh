#!/bin/bash
# Round 5, confirmation on the tree that is HEAD (python / docs / profiles only changed since r5_final2.sh and r5_final_b.sh): exactly what the
# driver runs at round end -- the GPU suite with -x, smoke(), the default bench
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5_confirm
mkdir -p $O
cd $R
timeout 420 python -m pytest tests/ -x -q -m gpu > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
echo "smoke rc=$?" >> $O/smoke.log
timeout 200 python bench.py --gpus 1 --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err
echo done > $O/finished
