#!/bin/bash
# How often does a process land in the fast state of the fused tile kernel?  The same bench command in separate processes.
set -o pipefail
for i in $(seq 1 ${1:-16}); do
timeout -k 10 100 python3 bench.py --no-cpu-baseline --no-eighth --steps 40 | python3 -c "import sys,json; j=json.loads(sys.stdin.readlines()[-1]); r=j['roofline']; print('run $i', r['frac'], r['launch_ms'], r.get('measured_copy_GBs'))" || exit 1
done
