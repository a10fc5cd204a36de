#!/bin/bash
set -o pipefail
for pad in 0 16384 24576 32768 40960 49152 57344; do
echo -n "pad $pad "; LVGPU_LIN_PAD=$pad timeout -k 10 100 python3 bench.py --workload ${1:-fill_translucent} --no-cpu-baseline --no-eighth --steps 60 | python3 tools/bench_summary.py || exit 1
done
