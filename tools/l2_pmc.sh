#!/bin/bash
# Run on the GPU box (gpurun): one `rocprofv3 --pmc` pass per counter group over tools/microbench_l2 (one launch per
# kernel), CSVs under gpurun_out/l2pmc_<group>/.  tools/l2_pmc_fold.py turns them into profiles/round2_sq_tcp_summary.json.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
BIN=$R/tools/microbench_l2
i=0
for set in \
  "TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_STREAMING_REQ_sum" \
  "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
  "TCC_BUSY_sum TCC_CYCLE_sum TCC_TAG_STALL_sum TCC_BUBBLE_sum" \
  "TCC_READ_SECTORS_sum TCC_WRITE_SECTORS_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" \
  "TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_64B_sum TCC_NORMAL_WRITEBACK_sum TCC_NORMAL_EVICT_sum" \
  "TCC_LATENCY_FIFO_FULL_sum TCC_SRC_FIFO_FULL_sum TCC_IB_STALL_sum TCC_IB_REQ_sum" \
  "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" \
  "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum" \
  "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" \
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
  "TCP_TCR_RDRET_STALL_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
  "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum" \
  "TA_DATA_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum" \
  "TD_TC_STALL_sum GRBM_GUI_ACTIVE GRBM_COUNT" \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
  "SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM" \
  "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL" ; do
  i=$((i+1))
  if [ $i -lt ${FIRST:-1} ]; then continue; fi
  # at most 4 counters of one block per pass (TA / TD / SQ groups of 5-8 abort rocprofv3 at start-up with error 38:
  # "Request exceeds the capabilities of the hardware", gpurun_out/l2pmc_12.log of round 2); every pass under a timeout
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/l2pmc_$i -- $BIN once > $R/gpurun_out/l2pmc_$i.log 2>&1 || echo "pmc group $i ($set) failed"
  echo "group $i done"
done
