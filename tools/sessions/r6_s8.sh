#!/bin/bash
# Round 6, session 8: two probes.  (1) tools/calib/overlap_probe.hip: can a CU run fp32 MFMAs and HBM streaming at full rate at once?
# (2) tools/pair_probe.py: the texel-pair load unit for 16-bit C = 16 features, loads only, against the single-texel probe and the product.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6_s8
mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 $R/tools/calib/overlap_probe.hip -o /tmp/overlap_probe > $O/overlap_build.log 2>&1
timeout 300 /tmp/overlap_probe > $O/overlap_probe.jsonl 2> $O/overlap_probe.err
cd $R
timeout 300 python tools/pair_probe.py --batch 2 > $O/pair_probe_b2.jsonl 2> $O/pair_probe_b2.err
timeout 300 python tools/pair_probe.py --batch 8 > $O/pair_probe_b8.jsonl 2> $O/pair_probe_b8.err
timeout 300 python tools/pair_probe.py --H 864 --W 1152 --src 7 --batch 4 > $O/pair_probe_cfg3.jsonl 2> $O/pair_probe_cfg3.err
echo done > $O/finished
