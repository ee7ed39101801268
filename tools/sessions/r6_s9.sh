#!/bin/bash
# Round 6, session 9: the overlap probe with other compute instructions (VALU, bf16 MFMA) and with the compute / memory waves on different CUs / XCDs
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6_s9
mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 $R/tools/calib/overlap_probe.hip -o /tmp/overlap_probe > $O/overlap_build.log 2>&1
timeout 400 /tmp/overlap_probe > $O/overlap_probe.jsonl 2> $O/overlap_probe.err
rocm-smi --showpower --showclocks > $O/smi_idle.txt 2>&1
echo done > $O/finished
