#!/bin/bash
# Address-translation counters of the fused tile kernel, the linear blend and the rotation alone (run on the GPU box):
# UTCL1 requests, hits and misses per launch.  tools/summarize_counters.py gpurun_out/prof_tlb prints them per kernel.
export TMPDIR=/tmp
OUT=gpurun_out/prof_tlb
for w in blend_flush blend rotate blend_flush_4k; do
  rm -rf $OUT/$w; mkdir -p $OUT/$w
  timeout -s KILL 120 rocprofv3 --pmc TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum --output-format csv -d $OUT/$w -- python3 bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-eighth > $OUT/$w.log 2>&1 || { echo "$w failed"; exit 1; }
  echo "$w done"
done
