#!/bin/bash
# Fuzzers on the final kernels (GPU box): single-descriptor entry points with the two flags, display lists.
set -o pipefail
mkdir -p gpurun_out/r3_fuzz
timeout -k 10 200 python -u tests/perf/fuzz_blend.py 150 71 > gpurun_out/r3_fuzz/blend_71.log 2>&1 || { tail -5 gpurun_out/r3_fuzz/blend_71.log; exit 1; }
tail -1 gpurun_out/r3_fuzz/blend_71.log
timeout -k 10 200 python -u tests/perf/fuzz_blend.py 150 72 > gpurun_out/r3_fuzz/blend_72.log 2>&1 || { tail -5 gpurun_out/r3_fuzz/blend_72.log; exit 1; }
tail -1 gpurun_out/r3_fuzz/blend_72.log
timeout -k 10 160 python -u tests/perf/fuzz_list.py 120 73 > gpurun_out/r3_fuzz/list_73.log 2>&1 || { tail -5 gpurun_out/r3_fuzz/list_73.log; exit 1; }
tail -1 gpurun_out/r3_fuzz/list_73.log
