#!/bin/bash
# Experiment (run on the GPU box): parity of the phased quarter-turn kernels for every rows-per-phase
# setting, then the rate sweep of tools/bench_phases.py.  Stops at the first failure.
set -o pipefail
O=gpurun_out/r3_phases
mkdir -p $O
T="timeout -k 10"
$T 300 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $O/tests_default.log 2>&1 || { tail -30 $O/tests_default.log; exit 1; }
tail -2 $O/tests_default.log
for nr in 4 2; do
  LVGPU_TILE_NR=$nr LVGPU_COL_NR=$nr LVGPU_ANY_NR=$nr $T 300 python -u -m pytest tests/test_gpu_flush.py tests/test_gpu_fullsize.py -m gpu -x -q --timeout 120 --timeout-method thread > $O/tests_nr$nr.log 2>&1 || { tail -30 $O/tests_nr$nr.log; exit 1; }
  tail -1 $O/tests_nr$nr.log
  LVGPU_TILE_NR=$nr LVGPU_COL_NR=$nr LVGPU_ANY_NR=$nr $T 120 python -u tests/perf/fuzz_blend.py 40 $((30 + nr)) > $O/fuzz_nr$nr.log 2>&1 || { tail -30 $O/fuzz_nr$nr.log; exit 1; }
  tail -2 $O/fuzz_nr$nr.log
done
$T 600 python -u tools/bench_phases.py > $O/bench_phases.txt 2>&1 || { tail -30 $O/bench_phases.txt; exit 1; }
echo "sweep done: $(wc -l < $O/bench_phases.txt) lines"
