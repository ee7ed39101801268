#!/bin/bash
# Round 6, session 24: conv0.1 of the fused FeatureNet stem in split-bf16 arithmetic: GPU parity tests (errors against fp64 printed), the step with
# it and with DMVS_STEM_EXACT=1, the kernel stats
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6_s24
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_ops.py tests/test_modules.py -x -q -m gpu -k "stem or feature" -s > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu.log
run() { name=$1; shift; env "$@" > $O/bench_$name.json 2> $O/bench_$name.err; }
run split_stem DMVS_X=0 timeout 300 python bench.py --no-batch-sweep --no-cpu-baseline --no-probe --steps 8 --warmup 2
run exact_stem DMVS_STEM_EXACT=1 timeout 300 python bench.py --no-batch-sweep --no-cpu-baseline --no-probe --steps 8 --warmup 2
run split_stem_b DMVS_X=0 timeout 300 python bench.py --no-batch-sweep --no-cpu-baseline --no-probe --steps 8 --warmup 2
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b96 -- python $R/bench.py --steps 5 --warmup 2 --no-batch-sweep --no-cpu-baseline --no-probe > $O/prof_b96_line.json 2> $O/prof_b96.err
cp $(find $O/prof_b96 -name "*kernel_stats.csv" | head -1) $O/b96_kernel_stats.csv 2>/dev/null
rm -rf $O/prof_b96
echo done > $O/finished
