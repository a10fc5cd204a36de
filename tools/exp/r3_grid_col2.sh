#!/bin/bash
# Heights with H % 64 == 32 below 512 rows: tile kernel against column kernel (GPU box).
set -o pipefail
for m in blend_flush rotate blend_flush_nowb; do
for s in "128 160 32768" "240 160 16384" "320 224 8192" "320 288 8192" "480 352 4096" "480 416 4096" "320 96 16384"; do
set -- $s
for k in tile col; do
echo -n "$m $1x$2 $k "; LVGPU_QT_KERNEL=$k timeout -k 10 100 python3 bench.py --workload $m --width $1 --height $2 --frames-per-gpu $3 --no-cpu-baseline --no-eighth --steps 30 | python3 tools/bench_summary.py | awk '{print $(NF-4), $(NF-3), $(NF-2)}' || exit 1
done; done; done
