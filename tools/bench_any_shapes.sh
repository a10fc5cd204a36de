#!/bin/bash
# The quarter-turn flush, alone and fused with the blend, over frame sizes off the 64-pixel grid (run on the GPU box):
# one line per size with the launcher's own choice of kernel and workgroup size, then, with SWEEP=1 in the environment,
# the same sizes with LVGPU_ANY_WPB = 2..8 wavefronts per workgroup of k_tile_any (the measurements its launch rule
# in vvc_amd/csrc/lvgpu_tiles.hip was derived from).  Output: gpurun_out/any_shapes.log, one bench_summary line each.
O=gpurun_out/any_shapes.log
mkdir -p gpurun_out
: > $O
run() { # label, environment assignment, bench arguments...
  local label=$1; shift; local envs=$1; shift
  local line
  line=$(env $envs timeout -k 10 120 python bench.py --no-cpu-baseline --no-eighth --steps 30 --warmup 3 "$@" 2>/dev/null | tail -n 1 | python tools/bench_summary.py) || return 1
  echo "$label | $envs | $line" >> $O
}
SIZES=("466 466 2048" "454 454 2048" "390 390 4096" "360 360 4096" "240 135 16384" "135 240 16384" "320 172 8192" "172 320 8192"
       "240 240 8192" "128 160 16384" "80 160 32768" "320 200 8192" "481 321 4096" "482 320 4096" "480 322 4096" "480 272 4096"
       "480 328 4096" "464 464 2048" "1024 600 1024" "1280 720 512" "1920 1080 256" "3840 2160 64"
       "640 480 2048" "800 480 2048" "720 720 1024" "1280 800 512" "2560 1440 128" "1023 600 1024" "1278 718 512")
for wl in rotate blend_flush; do
  for wh in "${SIZES[@]}"; do
    set -- $wh
    run "$wl ${1}x${2}" "WPB=auto" --workload $wl --width $1 --height $2 --frames-per-gpu $3 || exit 1
    if [ "$SWEEP" = 1 ]; then
      for v in 2 3 4 5 6 7 8; do
        run "$wl ${1}x${2}" "LVGPU_ANY_WPB=$v" --workload $wl --width $1 --height $2 --frames-per-gpu $3 || exit 1
      done
    fi
  done
done
cat $O
