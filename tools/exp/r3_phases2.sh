#!/bin/bash
# Experiment (GPU box): the whole GPU suite on the new launch rules, then both quarter-turn kernels on the
# sizes both take (tools/bench_phases.py both), the sweep on the launcher's own choices, and the bench lines.
set -o pipefail
O=gpurun_out/r3_phases2
mkdir -p $O
T="timeout -k 10"
$T 300 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
$T 500 python -u tools/bench_phases.py both > $O/bench_both.txt 2>&1 || { tail -30 $O/bench_both.txt; exit 1; }
$T 500 python -u tools/bench_phases.py > $O/bench_phases.txt 2>&1 || { tail -30 $O/bench_phases.txt; exit 1; }
for w in blend_flush blend_flush_nowb blend_flush_466 blend_flush_4k; do
  $T 200 python bench.py --workload $w --no-cpu-baseline --no-eighth > $O/bench_$w.json 2> $O/bench_$w.err || { cat $O/bench_$w.err; exit 1; }
  python tools/bench_summary.py < $O/bench_$w.json
done
