#!/bin/bash
# Round 6, session 15: SQ counters of the split-bf16 kernel on the 32 -> 32 3x3 layer at 96 x 128 x 160 (and the fp32 kernel beside it)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6_s15
mkdir -p $O
export CONV_2D_ONLY=1 CONV_ONLY="unet 32->32 3x3 1/4 B96"
for ar in split fp32; do
  CONV_ARITH=$ar timeout 120 python $R/tools/conv_bench.py > $O/time_$ar.jsonl 2>/dev/null
  CONV_ARITH=$ar timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/p1_$ar -- python $R/tools/conv_bench.py > /dev/null 2>&1
  CONV_ARITH=$ar timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD --output-format csv -d $O/p2_$ar -- python $R/tools/conv_bench.py > /dev/null 2>&1
  python $R/tools/pmc_kernel.py conv2d_mfma_kernel $O/p1_$ar $O/p2_$ar > $O/pmc_$ar.txt 2>&1
  rm -rf $O/p1_$ar $O/p2_$ar
done
echo done > $O/finished
