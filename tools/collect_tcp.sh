#!/bin/bash
# Requests between the vector L1 (TCP) and the L2 (TCC) of the main kernels, and how long reads stay outstanding
# (run on the GPU box): TCP_TCC_READ_REQ_LATENCY / kernel cycles = read requests in flight.
# tools/summarize_counters.py gpurun_out/prof_tcp prints them per kernel.
export TMPDIR=/tmp
OUT=gpurun_out/prof_tcp
PASSES=(
 "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum"
 "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_REQ_sum TCC_READ_sum"
 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
)
for w in ${@:-blend_flush blend rotate map_opaque fill_masked blend_flush_4k}; do
  for i in "${!PASSES[@]}"; do
    d=$OUT/$w/pass$i
    rm -rf $d; mkdir -p $d
    timeout -s KILL 120 rocprofv3 --pmc ${PASSES[$i]} --output-format csv -d $d -- python3 bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-eighth > $d.log 2>&1 || { echo "$w pass $i failed"; tail -n 3 $d.log; exit 1; }
  done
  echo "$w done"
done
python3 tools/summarize_counters.py $OUT > gpurun_out/prof_tcp_summary.txt
