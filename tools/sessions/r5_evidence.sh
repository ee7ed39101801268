#!/bin/bash
# Round 5, evidence only (no code under test changes): the L2 / L1 counters of GetCost on the 32 x 2-pixel tile mapping, same passes as
# profiles/r4_pmc_getcost_quad_b96.txt (the 64-pixel row segments); kernel stats of the batch-1 forward
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5_evidence
mkdir -p $O
cd /tmp
for pass in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"; do
  tag=$(echo $pass | cut -d' ' -f1)
  DIAG_ITERS=3 timeout 200 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $O/pmc_$tag -- python $R/tools/diag_r4.py getcost_pmc > $O/pmc_$tag.log 2>&1
  python $R/tools/pmc_kernel.py getcost_quad $O/pmc_$tag >> $O/getcost_pmc_b96.txt 2>> $O/pmc_reduce.err
  rm -rf $O/pmc_$tag
done
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b1 -- python $R/bench.py --batch 1 --steps 30 --warmup 5 --no-batch-sweep --no-cpu-baseline > $O/prof_b1.log 2>&1
cp $(find $O/prof_b1 -name "*kernel_stats.csv" | head -1) $O/b1_kernel_stats.csv 2>/dev/null
rm -rf $O/prof_b1
echo done > $O/finished
