#!/bin/bash
# Round 5, GPU session 4: soak -- the whole GPU suite twice more on a fresh box (an intermittent failure is what cost round 4 its run), the
# tile-walking stress tests with 120 launches each, smoke(), and the new eval / scene tests
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5_s4
mkdir -p $O
cd $R
timeout 600 python -m pytest tests -q -m gpu -x > $O/pytest_gpu_1.log 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu_1.log
timeout 600 python -m pytest tests -q -m gpu -x > $O/pytest_gpu_2.log 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu_2.log
DMVS_WALK_REPS=120 timeout 400 python -m pytest tests/test_ops.py -q -m gpu -k "reproducible" > $O/pytest_soak.log 2>&1
echo "pytest rc=$?" >> $O/pytest_soak.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
echo "smoke rc=$?" >> $O/smoke.log
echo done > $O/finished
