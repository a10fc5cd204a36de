#!/bin/bash
set -o pipefail
for pad in 0 4096 8192 12288 16384 20480 28672 36864 49152; do
LVGPU_ARGB_PAD=$pad timeout -k 10 120 python3 bench.py --workload fill_masked_argb --no-cpu-baseline --no-eighth --steps 60 | python3 -c "import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print('pad $pad', j['roofline']['frac'])" || exit 1
done
