#!/bin/bash
# Column kernel with two tiles per wavefront (GPU box): parity under the forced setting, then 1080p and neighbours.
set -o pipefail
LVGPU_COL_WPB=9 LVGPU_COL_TPW=2 timeout -k 10 400 python -u -m pytest tests/test_gpu_flush.py -x -q -m gpu --timeout 180 --timeout-method thread -k "rotate or strip or discard" 2>&1 | tail -3 || exit 1
run() { echo -n "$1 $2x$3 WPB=$5 TPW=$6 NR=$7 "; LVGPU_QT_KERNEL=col LVGPU_COL_WPB=$5 LVGPU_COL_TPW=$6 LVGPU_COL_NR=$7 timeout -k 10 100 python3 bench.py --workload $1 --width $2 --height $3 --frames-per-gpu $4 --no-cpu-baseline --no-eighth --steps 30 | python3 tools/bench_summary.py | awk '{print $(NF-4), $(NF-3), $(NF-2)}' || exit 1; }
for m in blend_flush rotate blend_flush_nowb; do
for nr in 4 2; do
run $m 1920 1080 256 9 1 $nr || exit 1
run $m 1920 1080 256 9 2 $nr || exit 1
run $m 1600 1200 256 10 1 $nr || exit 1
run $m 1600 1200 256 10 2 $nr || exit 1
run $m 1280 720 512 12 1 $nr || exit 1
run $m 1280 720 512 6 2 $nr || exit 1
done; done
