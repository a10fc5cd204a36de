#!/bin/bash
# Run on the GPU box (through gpurun) from the repository root: rocprofv3 kernel-trace statistics and the two
# PMC passes (FETCH_SIZE, WRITE_SIZE -- separate runs, they do not fit one pass) for each bench workload.
# Output goes to gpurun_out/prof/<workload>/{kt,fetch,write}; tools/summarize_profiles.py turns it into profiles/.
set -e
export TMPDIR=/tmp
OUT=gpurun_out/prof
for w in blend_flush blend rotate; do
  mkdir -p $OUT/$w
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$w/kt -- python3 bench.py --workload $w --no-cpu-baseline --no-eighth > $OUT/$w/kt.log 2>&1
  timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/$w/fetch -- python3 bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-eighth > $OUT/$w/fetch.log 2>&1
  timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/$w/write -- python3 bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-eighth > $OUT/$w/write.log 2>&1
  echo "$w done"
done
