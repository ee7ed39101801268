#!/bin/bash
# Round 5, GPU session 2: after the LDS-DMA barrier fix (explicit vmcnt(0) before every barrier that publishes DMA'd LDS data) and with the
# paired 3-D kernels as the default: the whole GPU suite (new: tile-walking stress tests, scene mode), the reproducibility trace again,
# the GetCost ceiling probe (gcexp4 builds), the bench line with the conv table, the scene-mode line
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5_s2
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout 300 python tools/determinism.py cfg3 --runs 8 > $O/determinism_cfg3.jsonl 2> $O/determinism_cfg3.err
timeout 300 python tools/determinism.py cfg5 --runs 6 > $O/determinism_cfg5.jsonl 2> $O/determinism_cfg5.err
timeout 200 python tools/diag_r4.py getcost > $O/getcost_probe.jsonl 2> $O/getcost_probe.err
timeout 100 python tools/diag_r4.py pair3d > $O/pair3d.jsonl 2> $O/pair3d.err
timeout 200 python bench.py --no-batch-sweep --no-cpu-baseline --steps 10 --warmup 2 --conv-table > $O/bench_default.json 2> $O/bench_default.err
timeout 200 python bench.py --scene-mode --steps 8 --warmup 2 > $O/bench_scene.json 2> $O/bench_scene.err
echo done > $O/finished
