#!/bin/bash
# Heights ON the 32-pixel grid: the tile kernel against the column kernel with a workgroup per column (GPU box).
set -o pipefail
for m in blend_flush rotate blend_flush_nowb; do
for s in "480 320 4096" "640 480 2048" "800 480 2048" "480 800 2048" "1024 768 1024" "1280 800 512" "1280 1024 512" "1920 1088 256" "2560 1440 128"; do
set -- $s
for k in tile col; do
echo -n "$m $1x$2 $k "; LVGPU_QT_KERNEL=$k timeout -k 10 100 python3 bench.py --workload $m --width $1 --height $2 --frames-per-gpu $3 --no-cpu-baseline --no-eighth --steps 30 | python3 tools/bench_summary.py | awk '{print $(NF-4), $(NF-3), $(NF-2)}' || exit 1
done; done; done
