#!/bin/bash
# Round 6, session 4: the weight-gradient kernel with 16-byte staging pieces + gradients added straight into the trainer's flat bucket: GPU
# tests of the training path, the cfg4 line and its kernel profile.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6_s4
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_ops.py tests/test_train.py tests/test_trainer.py tests/test_modules.py -x -q -m gpu -k "wgrad or autograd or backward or training_step or full_step or cfg4 or checkpoint or adamw or plane_sweep or warp_corr_init" > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout 400 python bench.py --config cfg4 --steps 10 --warmup 3 > $O/bench_cfg4.json 2> $O/bench_cfg4.err
DMVS_CONV_V16=0 timeout 400 python bench.py --config cfg4 --steps 10 --warmup 3 > $O/bench_cfg4_pieces4.json 2> $O/bench_cfg4_pieces4.err
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg4 -- python $R/bench.py --config cfg4 --steps 4 --warmup 2 > $O/prof_cfg4_line.json 2> $O/prof_cfg4.err
cp $(find $O/prof_cfg4 -name "*kernel_stats.csv" | head -1) $O/cfg4_kernel_stats.csv 2>/dev/null
rm -rf $O/prof_cfg4
echo done > $O/finished
