#!/bin/bash
# Round 6, session 3: GPU tests of the interleaved backward as the default + the split plane sweep; the plane sweep A/B in the step; the kernel
# profile of the cfg4 training step after the backward change.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6_s3
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_ops.py tests/test_train.py tests/test_trainer.py -x -q -m gpu -k "warp_corr_init or plane_sweep or backward or training_step or full_step or getcost" > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu.log
for m in band split band split; do
  DMVS_PLANE_SWEEP=$m timeout 300 python bench.py --no-batch-sweep --no-cpu-baseline --no-probe --steps 10 --warmup 3 > $O/bench_sweep_$m.$RANDOM.json 2>> $O/bench_sweep.err
done
timeout 400 python bench.py --config cfg4 --steps 10 --warmup 3 > $O/bench_cfg4.json 2> $O/bench_cfg4.err
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg4 -- python $R/bench.py --config cfg4 --steps 4 --warmup 2 > $O/prof_cfg4_line.json 2> $O/prof_cfg4.err
cp $(find $O/prof_cfg4 -name "*kernel_stats.csv" | head -1) $O/cfg4_kernel_stats.csv 2>/dev/null
rm -rf $O/prof_cfg4
echo done > $O/finished
