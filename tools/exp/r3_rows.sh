#!/bin/bash
# Round-3 experiment 3 (GPU box): whole-piece row writes (LVGPU_BLEND_FLAG_OWNS_ROWS) -- parity, then the clipped-area
# workloads with and without the flag and for every piece size, then the new C host example's test.
set -o pipefail
O=gpurun_out/r3_rows
mkdir -p $O
T="timeout -k 10"
$T 300 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
: > $O/rows.txt
for w in clipped_small clipped_large; do
  $T 100 python bench.py --workload $w --no-cpu-baseline --no-eighth | python tools/bench_summary.py | sed "s/^/$w plain | /" >> $O/rows.txt || exit 1
  for sp in 8 16 32 64; do
    LVGPU_ROWS_SPAN=$sp $T 100 python bench.py --workload ${w}_owned --no-cpu-baseline --no-eighth | python tools/bench_summary.py | sed "s/^/$w span $sp | /" >> $O/rows.txt || exit 1
  done
done
cat $O/rows.txt
