#!/bin/bash
# Counters of the quarter-turn flush on frames whose rows (482x320) or panel rows (480x322) are not 16-byte aligned,
# beside aligned frames through the same kernel (480x200) and through k_tile (480x320).  Run on the GPU box;
# tools/summarize_counters.py gpurun_out/prof_mis prints them per shape.
export TMPDIR=/tmp
OUT=gpurun_out/prof_mis
PASSES=(
 "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"
 "TCC_HIT_sum TCC_MISS_sum TCC_READ_sum TCC_WRITE_sum"
 "TCC_READ_SECTORS_sum TCC_WRITE_SECTORS_sum TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum"
 "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TD_TC_STALL_sum TD_TD_BUSY_sum"
 "TCP_TA_TCP_STATE_READ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_WAVEFRONTS_sum"
 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAVE_CYCLES"
 "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_ACTIVE_INST_VALU"
)
WL=${WL:-rotate}
for wh in ${@:-480x200 482x320 480x322 480x320}; do
  W=${wh%x*}; H=${wh#*x}
  for i in "${!PASSES[@]}"; do
    d=$OUT/${WL}_$wh/pass$i
    rm -rf $d; mkdir -p $d
    timeout -s KILL 120 rocprofv3 --pmc ${PASSES[$i]} --output-format csv -d $d -- python3 bench.py --workload $WL --width $W --height $H --frames-per-gpu 2048 --steps 3 --warmup 1 --no-cpu-baseline --no-eighth > $d.log 2>&1 || { echo "$wh pass $i failed"; tail -n 3 $d.log; exit 1; }
  done
  echo "$wh done"
done
python3 tools/summarize_counters.py $OUT > gpurun_out/prof_mis_summary.txt
