#!/bin/bash
# Which of the full-size tests leaves the device in the fast state?  (GPU box)
set -o pipefail
b() { timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-eighth 2>/dev/null | python3 -c "import sys,json; j=[l for l in sys.stdin if l.startswith('{\"metric')][-1]; r=json.loads(j)['roofline']; print('  bench', r['frac'], r['launch_ms'])" || exit 1; }
for t in test_config2_blend_4096_frames test_config3_fused_blend_rotate_4096_frames test_rotation_round_trips_4096_frames test_fill_properties_4096_frames test_config5_large_canvas_3840x2160 test_two_host_threads_on_their_own_streams test_results_do_not_depend_on_the_streams_cu_mask test_config4_65536_frames_and_their_shards; do
echo "after $t:"
timeout -k 10 400 python -u -m pytest tests/test_gpu_fullsize.py -x -q -m gpu --timeout 180 --timeout-method thread -k $t > /dev/null 2>&1; b; b
done
