#!/bin/bash
# The launch parameters of the any-size and column kernels were chosen with the buffers in the allocator's placement:
# the same sweeps with the buffers placed by region (GPU box).  One line per run.
set -o pipefail
MAP=$(timeout -k 10 120 python3 -m vvc_amd.placement 2>/dev/null | tail -n 1) || exit 1
echo "region map $MAP"
run() { # label, env assignment, bench args
  local label=$1; shift; local envs=$1; shift
  echo -n "$label | $envs | "
  env $envs timeout -k 10 120 python3 bench.py --no-cpu-baseline --no-eighth --steps 30 --warmup 3 --region-map $MAP "$@" 2>/dev/null | tail -n 1 | python3 tools/bench_summary.py | awk '{print $(NF-3), $(NF-2)}' || exit 1
}
for wh in "466 466 2048" "390 390 4096" "481 321 4096" "172 320 8192" "1023 600 1024" "1278 718 512" "240 135 16384"; do
  set -- $wh
  for e in WPB=auto LVGPU_ANY_WPB=4 LVGPU_ANY_WPB=5 LVGPU_ANY_WPB=6 LVGPU_ANY_WPB=8 LVGPU_ANY_WPB=10 LVGPU_ANY_WPB=12 LVGPU_ANY_WPB=16 LVGPU_ANY_NR=4 LVGPU_ANY_NR=8 LVGPU_QT_KERNEL=col; do
    run "blend_flush ${1}x${2}" $e --workload blend_flush --width $1 --height $2 --frames-per-gpu $3 || exit 1
  done
done
for wh in "3840 2160 64" "1920 1080 256" "1024 600 1024" "480 272 4096"; do
  set -- $wh
  for e in WPB=auto LVGPU_COL_WPB=8 LVGPU_COL_WPB=12 LVGPU_COL_WPB=16 LVGPU_COL_NR=2 LVGPU_COL_NR=8 LVGPU_COL_TPW=1 LVGPU_COL_TPW=2; do
    run "blend_flush ${1}x${2}" $e --workload blend_flush --width $1 --height $2 --frames-per-gpu $3 || exit 1
  done
done
