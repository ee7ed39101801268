#!/bin/bash
# (record: gcc<threads>s<shift> = `-DDMVS_GC_BLOCK=<threads> -DDMVS_GC_TW_SHIFT=<shift>` builds; results: profiles/r5_getcost_mapping_sweep_b96.jsonl)
# Round 5, GPU session 8: GetCost pixel mapping x workgroup size, every combination as a variant build with a compile-time tile width
# (gcc<threads>s<log2 tile width>; s6 at 256 threads = the old 64-pixel row segment) next to the product (runtime tile width)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5_s8
mkdir -p $O
cd $R
timeout 400 python tools/diag_r4.py getcost > $O/getcost_map.jsonl 2> $O/getcost_map.err
echo done > $O/finished
