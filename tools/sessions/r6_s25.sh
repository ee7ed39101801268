#!/bin/bash
# Round 6, session 25 (no code change under test: evidence for DESIGN 4.5): SQ counters of the FINAL split-bf16 kernel on the 32 -> 32 3x3 layer at
# 96 x 128 x 160 and on 64 -> 64 at 576 x 64 x 80, the exact-fp32 kernels beside them
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6_s25
mkdir -p $O
export CONV_2D_ONLY=1
for layer in "unet 32->32 3x3 1/4 B96" "feat 64->64 3x3 1/8 N576"; do
  tag=$(echo "$layer" | tr -c 'a-zA-Z0-9' '_' | cut -c1-20)
  export CONV_ONLY="$layer"
  for ar in split fp32; do
    CONV_ARITH=$ar DMVS_CONV_SPLIT_ALL=1 timeout 120 python $R/tools/conv_bench.py > $O/time_${tag}_$ar.jsonl 2>/dev/null
    CONV_ARITH=$ar timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/p1 -- python $R/tools/conv_bench.py > /dev/null 2>&1
    CONV_ARITH=$ar timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD --output-format csv -d $O/p2 -- python $R/tools/conv_bench.py > /dev/null 2>&1
    python $R/tools/pmc_kernel.py conv2d_mfma_kernel $O/p1 $O/p2 > $O/pmc_${tag}_$ar.txt 2>&1
    rm -rf $O/p1 $O/p2
  done
done
echo done > $O/finished
