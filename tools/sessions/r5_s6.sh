#!/bin/bash
# (record: the gctile* / gcexp4tile* variant names belong to the first, macro-only form of the mapping, `-DDMVS_GC_TILE_W=n`; superseded by DMVS_GC_BLOCK / DMVS_GC_TW_SHIFT)
# Round 5, GPU session 6: GetCost with the workgroup's 64 pixels as a 2-D tile (32x2 / 16x4 / 8x8) instead of a 64-pixel row segment -- product
# arithmetic and the ceiling probe in each mapping (variant builds, tools/build_variant.py ... -DDMVS_GC_TILE_W=n)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5_s6
mkdir -p $O
cd $R
timeout 300 python tools/diag_r4.py getcost > $O/getcost_tiles.jsonl 2> $O/getcost_tiles.err
echo done > $O/finished
