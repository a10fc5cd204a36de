#!/bin/bash
# L2 <-> memory request counters of the fused tile kernel, the linear blend and the rotation alone (run on the GPU
# box): request sizes, stalls on write / DRAM credits, hit rates.  Four TCC counters fit one pass.
# tools/summarize_counters.py gpurun_out/prof_l2 prints them per kernel.
export TMPDIR=/tmp
OUT=gpurun_out/prof_l2
PASSES=(
 "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"
 "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum"
 "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum TCC_BUSY_sum"
 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_CYCLE_sum"
)
for w in ${@:-blend_flush blend rotate}; do
  for i in "${!PASSES[@]}"; do
    d=$OUT/$w/pass$i
    rm -rf $d; mkdir -p $d
    timeout -s KILL 120 rocprofv3 --pmc ${PASSES[$i]} --output-format csv -d $d -- python3 bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-eighth > $d.log 2>&1 || { echo "$w pass $i failed"; tail -3 $d.log; exit 1; }
  done
  echo "$w done"
done
