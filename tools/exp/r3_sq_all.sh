#!/bin/bash
# VALU issue share of the dominant kernel of several workloads (GPU box).
set -o pipefail
export TMPDIR=/tmp
for wl in clipped_large_argb clipped_small_argb fill_masked_argb clipped_large; do
OUT=gpurun_out/prof_sq/$wl
mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT -- python3 bench.py --workload $wl --no-cpu-baseline --no-eighth --steps 10 --warmup 3 > $OUT/log.txt 2>&1 || { tail -5 $OUT/log.txt; exit 1; }
python3 - $wl <<'PY'
import csv, glob, collections, sys
wl = sys.argv[1]
f = sorted(glob.glob("gpurun_out/prof_sq/%s/**/*counter_collection.csv" % wl, recursive=True))[-1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    acc[r["Kernel_Name"].split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
best = max(acc.items(), key=lambda kv: sum(kv[1].get("GRBM_GUI_ACTIVE", [0])))
k, c = best
m = {n: sum(v) / len(v) for n, v in c.items()}
busy = m["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / m["GRBM_GUI_ACTIVE"]
print("%-18s %-50s VALU %.0fM insts, active/(gui cycles) %.2f, gui %.0f cycles" % (wl, k, m["SQ_INSTS_VALU"] / 1e6, busy, m["GRBM_GUI_ACTIVE"]))
PY
done
