#!/bin/bash
# ARGB8565 destination after a kernel change (GPU box): its tests, then the rates.
set -o pipefail
mkdir -p gpurun_out/r3_argb
timeout -k 10 400 python -u -m pytest tests/test_gpu_blend.py tests/test_gpu_list.py tests/test_gpu_lvgl_dropin.py -x -q -m gpu --timeout 180 --timeout-method thread -k "argb or transp or alpha or list or dropin or aa" > gpurun_out/r3_argb/tests.log 2>&1 || { tail -30 gpurun_out/r3_argb/tests.log; exit 1; }
tail -1 gpurun_out/r3_argb/tests.log
for i in 1 2; do
timeout -k 10 120 python3 bench.py --workload fill_masked_argb --no-cpu-baseline --no-eighth --steps 60 | python3 -c "import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print('fill_masked_argb', j['roofline']['frac'], j['roofline'].get('kernel'))" || exit 1
done
timeout -k 10 200 python3 tools/bench_modes.py 2>&1 | tail -12
