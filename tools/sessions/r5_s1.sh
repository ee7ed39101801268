#!/bin/bash
# (record of the run at commit 138e93a: the diag modes heavymt / pairwreg / pair8 were folded into `pair3d` afterwards, when the paired kernels became the default and the other opt-ins were deleted)
# Round 5, GPU session 1: (1) where the cfg5 run-to-run difference of the round-4 driver run comes from (tools/determinism.py), (2) the
# whole GPU suite in its new order, WITHOUT -x, so that every red test is seen at once, (3) the opt-in kernel instantiations round 4 left
# untimed, each per layer with its bit-identity check (they become the default or are deleted after this session).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5_s1
mkdir -p $O
cd $R
timeout 420 python tools/determinism.py cfg5 --runs 4 > $O/determinism_cfg5.jsonl 2> $O/determinism_cfg5.err
timeout 200 python tools/determinism.py cfg3 --runs 3 > $O/determinism_cfg3.jsonl 2> $O/determinism_cfg3.err
timeout 600 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu.log
for m in heavymt pairwreg pair8 convexp; do timeout 100 python tools/diag_r4.py $m > $O/$m.jsonl 2> $O/$m.err; done
DMVS_CONV3D_PAIR_WREG=1 DMVS_CONV3D_PAIR8=1 timeout 150 python bench.py --no-batch-sweep --no-cpu-baseline --steps 10 --warmup 2 > $O/bench_pairwreg.json 2> $O/bench_pairwreg.err
timeout 150 python bench.py --no-batch-sweep --no-cpu-baseline --steps 10 --warmup 2 --conv-table > $O/bench_default.json 2> $O/bench_default.err
echo done > $O/finished
