#!/bin/bash
# The any-size quarter-turn kernel with 9 to 16 wavefronts per workgroup on frames of 513 rows and more (GPU box).
set -o pipefail
run() {  # mode W H frames nr wpb
  echo -n "$1 $2x$3 NR=$5 WPB=$6 "
  LVGPU_QT_KERNEL=any LVGPU_ANY_NR=$5 LVGPU_ANY_WPB=$6 timeout -k 10 100 python3 bench.py --workload $1 --width $2 --height $3 --frames-per-gpu $4 --no-cpu-baseline --no-eighth --steps 30 | python3 tools/bench_summary.py | awk '{print $(NF-4), $(NF-3), $(NF-2)}' || exit 1
}
for mode in blend_flush rotate blend_flush_nowb; do
for nr in 4 2; do
run $mode 1024 600 1024 $nr 10 || exit 1
run $mode 1024 600 1024 $nr 5 || exit 1
run $mode 1280 720 512 $nr 12 || exit 1
run $mode 1280 720 512 $nr 6 || exit 1
run $mode 1920 1080 256 $nr 9 || exit 1
run $mode 1920 1080 256 $nr 12 || exit 1
run $mode 1920 1080 256 $nr 16 || exit 1
run $mode 3840 2160 64 $nr 9 || exit 1
run $mode 3840 2160 64 $nr 12 || exit 1
run $mode 3840 2160 64 $nr 16 || exit 1
run $mode 800 480 2048 $nr 8 || exit 1
run $mode 720 720 1024 $nr 12 || exit 1
done; done
