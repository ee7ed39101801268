#!/bin/bash
# (record: at this commit the tile width was a kernel argument; gcblk* = `-DDMVS_GC_BLOCK=n` builds)
# Round 5, GPU session 7: GetCost on 32 x 2-pixel tiles (now the product mapping): 512-thread workgroups (32 x 4 tiles) and 128-thread ones
# (32 x 1) against the 256-thread product, and the ceiling probe in the new mapping; the getcost parity tests on the GPU
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5_s7
mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_ops.py tests/test_modules.py -q -m gpu -k "getcost or get_cost" > $O/pytest_getcost.log 2>&1
echo "pytest rc=$?" >> $O/pytest_getcost.log
timeout 300 python tools/diag_r4.py getcost > $O/getcost_blocks.jsonl 2> $O/getcost_blocks.err
echo done > $O/finished
