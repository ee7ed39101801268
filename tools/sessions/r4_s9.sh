#!/bin/bash
# Round 4, GPU session 9: SepConvGRU without the LDS gating pass (out_mul), the 16-byte 1x1 form with batched residual loads / 2 n-tiles
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/s9
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_ops.py tests/test_model_gpu.py tests/test_modules.py -q -m gpu -k "conv2d or golden or gru or a10 or a11 or a9" > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout 400 python bench.py --no-batch-sweep --no-cpu-baseline --steps 10 --warmup 2 --conv-table > $O/bench.json 2> $O/bench.err
DMVS_LIB=$R/tools/calib/libdmvs_hip_px4nt2.so timeout 400 python bench.py --no-batch-sweep --no-cpu-baseline --steps 10 --warmup 2 --conv-table > $O/bench_px4nt2.json 2> $O/bench_px4nt2.err
echo done > $O/finished
