#!/bin/bash
# What before a process puts the default bench into the fast state?  (GPU box)
set -o pipefail
b() { timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-eighth 2>/dev/null | python3 -c "import sys,json; j=[l for l in sys.stdin if l.startswith('{\"metric')][-1]; r=json.loads(j)['roofline']; print('  bench', r['frac'], r['launch_ms'], r.get('measured_copy_GBs'))" || exit 1; }
echo "fresh box:"; b; b
echo "after a process that allocated, filled and freed 120 GB:"
timeout -k 10 200 python3 -c "
import torch
x=[torch.empty(10<<30, dtype=torch.uint8, device='cuda').fill_(1) for _ in range(12)]
torch.cuda.synchronize()
" ; b; b
echo "after smoke:"
timeout -k 10 200 python3 -c "import __graft_entry__ as g; g.smoke()" > /dev/null 2>&1; b; b
echo "after the full-size tests:"
timeout -k 10 400 python -u -m pytest tests/test_gpu_fullsize.py -x -q -m gpu --timeout 180 --timeout-method thread > /dev/null 2>&1; b; b
echo "after the whole suite:"
timeout -k 10 600 python -u -m pytest tests -x -q -m gpu --timeout 180 --timeout-method thread > /dev/null 2>&1; b; b
echo "after 20 s of sleep:"
sleep 20; b; b
