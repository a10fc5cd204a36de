#!/bin/bash
# Fuzzers on the final kernels (GPU box): single-descriptor entry points (both destination formats, the two flags),
# display lists.
set -o pipefail
mkdir -p gpurun_out/r3_fuzz
timeout -k 10 360 python -u tests/perf/fuzz_blend.py 300 ${1:-91} > gpurun_out/r3_fuzz/blend_a.log 2>&1 || { tail -5 gpurun_out/r3_fuzz/blend_a.log; exit 1; }
tail -1 gpurun_out/r3_fuzz/blend_a.log
timeout -k 10 360 python -u tests/perf/fuzz_blend.py 300 ${2:-92} > gpurun_out/r3_fuzz/blend_b.log 2>&1 || { tail -5 gpurun_out/r3_fuzz/blend_b.log; exit 1; }
tail -1 gpurun_out/r3_fuzz/blend_b.log
timeout -k 10 300 python -u tests/perf/fuzz_list.py 240 ${3:-93} > gpurun_out/r3_fuzz/list_c.log 2>&1 || { tail -5 gpurun_out/r3_fuzz/list_c.log; exit 1; }
tail -1 gpurun_out/r3_fuzz/list_c.log
