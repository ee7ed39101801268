#!/bin/bash
# Prepared at the end of round 4 (no GPU budget left) for the first GPU session of the next round: the experiment instantiations that
# exist but have never been timed, each per layer with its bit-identity check, and the two same-box step A/Bs they point at.
#   mtsweep   every multi-tap conv2d shape of the step with each tile shape forced, against the dispatcher's choice
#   heavymt   16 x 16 tiles for the stride-2 / 5x5 / 7x7 conv families (DMVS_TUNE_TILE_MT(4))
#   pairwreg  the 4 -> 8 paired 3-D kernel with its weights in registers (DMVS_TUNE3D_PAIR_WREG)
#   pair8     CostRegNet conv1 (8 -> 8) on a two-chunk paired kernel (DMVS_TUNE3D_PAIR8)
#   convexp   tall tiles again, now with the 16 x 64 form beside them (DMVS_TUNE_TALL(3))
#   stem      16-byte against 4-byte halo pieces, on random data and inside the model's step (open)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5_s1
mkdir -p $O
cd $R
for m in mtsweep heavymt pairwreg pair8 convexp stem; do timeout 300 python tools/diag_r4.py $m > $O/$m.jsonl 2> $O/$m.err; done
timeout 300 python bench.py --no-batch-sweep --no-cpu-baseline --steps 10 --warmup 2 --conv-table > $O/bench_default.json 2> $O/bench_default.err
DMVS_STEM_V16=0 timeout 300 python bench.py --no-batch-sweep --no-cpu-baseline --steps 10 --warmup 2 > $O/bench_stem4.json 2> $O/bench_stem4.err
DMVS_CONV3D_PAIR_WREG=1 DMVS_CONV3D_PAIR8=1 timeout 300 python bench.py --no-batch-sweep --no-cpu-baseline --steps 10 --warmup 2 > $O/bench_pairwreg.json 2> $O/bench_pairwreg.err
echo done > $O/finished
