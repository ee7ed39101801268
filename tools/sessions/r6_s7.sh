#!/bin/bash
# Round 6, session 7: the XCD-grouped tile order as the default of the tiled conv2d kernels, and the same order in the fused stem and the tiled
# 3-D kernels: GPU tests of the touched kernels, then the step with each family's grouping switched off in turn.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6_s7
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_ops.py -x -q -m gpu -k "xcd_grouped or conv3d or stem or conv2d" > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu.log
run() { name=$1; shift; env "$@" timeout 300 python bench.py --no-batch-sweep --no-cpu-baseline --no-probe --steps 8 --warmup 2 > $O/bench_$name.json 2> $O/bench_$name.err; }
run default_a DMVS_X=0
run all_off DMVS_CONV_XCD=1 DMVS_STEM_XCD=1 DMVS_CONV3D_XCD=1
run stem_off DMVS_STEM_XCD=1
run conv3d_off DMVS_CONV3D_XCD=1
run default_b DMVS_X=0
run stem8_conv3d8 DMVS_STEM_XCD=4 DMVS_CONV3D_XCD=4
run stem2_conv3d2 DMVS_STEM_XCD=2 DMVS_CONV3D_XCD=2
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b96 -- python $R/bench.py --steps 5 --warmup 2 --no-batch-sweep --no-cpu-baseline --no-probe > $O/prof_b96_line.json 2> $O/prof_b96.err
cp $(find $O/prof_b96 -name "*kernel_stats.csv" | head -1) $O/b96_kernel_stats.csv 2>/dev/null
rm -rf $O/prof_b96
echo done > $O/finished
