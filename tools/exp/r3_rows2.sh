#!/bin/bash
set -o pipefail
O=gpurun_out/r3_rows2
mkdir -p $O
T="timeout -k 10"
for r in 0 1; do
LVGPU_ROWS_REMAP=$r $T 300 python -u -m pytest tests/test_gpu_blend.py tests/test_gpu_extremes.py -m gpu -x -q --timeout 120 --timeout-method thread > $O/tests$r.log 2>&1 || { tail -40 $O/tests$r.log; exit 1; }
tail -1 $O/tests$r.log
done
: > $O/rows.txt
for w in clipped_small clipped_large clipped_small_owned clipped_large_owned; do
  for r in 0 1; do
  LVGPU_ROWS_REMAP=$r $T 100 python bench.py --workload $w --no-cpu-baseline --no-eighth | python tools/bench_summary.py | sed "s/^/$w remap $r | /" >> $O/rows.txt || exit 1
  done
done
cat $O/rows.txt
