#!/bin/bash
set -o pipefail
for pad in 0 12288 16384; do
echo "pad $pad"
LVGPU_ARGB_PAD=$pad timeout -k 10 200 python3 tools/bench_modes.py 2>&1 | grep "k_blend_argb_linear" || exit 1
done
