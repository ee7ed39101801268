#!/bin/bash
# Round 4, GPU session 1: diagnostics (what bounds GetCost), the opt-ins round 3 left untimed, the LDS-DMA probe.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/s1
mkdir -p $O
cd $R
timeout 900 python tools/diag_r4.py getcost > $O/getcost_diag.jsonl 2> $O/getcost_diag.err
timeout 300 python tools/diag_r4.py warp_init > $O/warp_init_diag.jsonl 2> $O/warp_init_diag.err
timeout 600 python tools/diag_r4.py optins > $O/optins.jsonl 2> $O/optins.err
(/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/calib/dma_probe.hip -o /tmp/dma_probe && timeout 120 /tmp/dma_probe) > $O/dma_probe.jsonl 2> $O/dma_probe.err
rocprofv3 -L > $O/counters_list.txt 2>&1
cd /tmp
for pass in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
  tag=$(echo $pass | cut -d' ' -f1)
  DIAG_ITERS=3 timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $O/pmc_$tag -- python $R/tools/diag_r4.py getcost_pmc > $O/pmc_$tag.log 2>&1
  python $R/tools/pmc_kernel.py getcost_quad $O/pmc_$tag >> $O/getcost_pmc_b96.txt 2>> $O/pmc_reduce.err
  rm -rf $O/pmc_$tag
done
echo done > $O/finished
