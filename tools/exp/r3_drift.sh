#!/bin/bash
set -o pipefail
O=gpurun_out/r3_drift
mkdir -p $O
for s in "480 328 4096" "480 320 4096" "466 466 2048"; do
  timeout -k 10 120 python -u tools/probe_drift.py $s > "$O/drift_${s// /_}.txt" 2>&1 || { tail -20 "$O/drift_${s// /_}.txt"; exit 1; }
  cat "$O/drift_${s// /_}.txt"
done
rocm-smi --showclocks --showpower 2>&1 | head -40 > $O/smi.txt || true
