#!/bin/bash
# Issue counters of the ARGB8565 masked fill (GPU box).
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/prof_argb_sq
mkdir -p $OUT
ARGS="bench.py --workload fill_masked_argb --no-cpu-baseline --no-eighth --steps 10 --warmup 3"
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT/sq -- python3 $ARGS > $OUT/sq.log 2>&1 || { tail -5 $OUT/sq.log; exit 1; }
python3 - <<'PY'
import csv, glob, collections
f = sorted(glob.glob("gpurun_out/prof_argb_sq/sq/**/*counter_collection.csv", recursive=True))[-1]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "k_blend_argb_linear" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(k, len(v), sum(v) / len(v))
PY
