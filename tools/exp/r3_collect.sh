#!/bin/bash
# Collection (GPU box): the profile set (tools/collect_profiles.sh), the display-list profile, the off-grid
# shape table, and two fuzz runs on the kernels as they are.
set -o pipefail
mkdir -p gpurun_out
bash tools/collect_profiles.sh > gpurun_out/collect.log 2>&1 || { tail -5 gpurun_out/collect.log; exit 1; }
tail -3 gpurun_out/collect.log
bash tools/collect_list_profile.sh > gpurun_out/collect_list.log 2>&1 || { tail -5 gpurun_out/collect_list.log; exit 1; }
bash tools/bench_any_shapes.sh > /dev/null 2>&1 || { tail -5 gpurun_out/any_shapes.log; exit 1; }
tail -3 gpurun_out/any_shapes.log
timeout -k 10 200 python -u tests/perf/fuzz_blend.py 120 301 > gpurun_out/fuzz_blend_r3.log 2>&1 || { tail -20 gpurun_out/fuzz_blend_r3.log; exit 1; }
tail -1 gpurun_out/fuzz_blend_r3.log
timeout -k 10 200 python -u tests/perf/fuzz_list.py 90 302 > gpurun_out/fuzz_list_r3.log 2>&1 || { tail -20 gpurun_out/fuzz_list_r3.log; exit 1; }
tail -1 gpurun_out/fuzz_list_r3.log
