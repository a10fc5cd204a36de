#!/bin/bash
# Collection (GPU box): the GPU suite, the profile set, the display-list profile, the off-grid shape table.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/gputests.log 2>&1 || { tail -40 gpurun_out/gputests.log; exit 1; }
tail -1 gpurun_out/gputests.log
bash tools/collect_profiles.sh > gpurun_out/collect.log 2>&1 || { tail -5 gpurun_out/collect.log; exit 1; }
tail -2 gpurun_out/collect.log
bash tools/collect_list_profile.sh > gpurun_out/collect_list.log 2>&1 || { tail -5 gpurun_out/collect_list.log; exit 1; }
bash tools/bench_any_shapes.sh > /dev/null 2>&1 || { tail -5 gpurun_out/any_shapes.log; exit 1; }
tail -2 gpurun_out/any_shapes.log
