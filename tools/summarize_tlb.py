"""Per kernel: mean UTCL1 requests / hits / misses per launch from gpurun_out/prof_tlb/<workload>/**/_counter_collection.csv."""
import csv
import glob
import sys
from collections import defaultdict

for w in sys.argv[1:] or ["blend_flush", "blend", "rotate", "blend_flush_4k"]:
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob("gpurun_out/prof_tlb/%s/**/*counter_collection.csv" % w, recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in acc.items():
        if "k_synth" in k or "rocclr" in k:
            continue
        m = {n: sum(v) / len(v) for n, v in c.items()}
        req = m.get("TCP_UTCL1_REQUEST_sum", 0)
        print("%-16s %-40s requests %12.0f  hits %12.0f  misses %10.0f  miss rate %.4f" % (
            w, k[-40:], req, m.get("TCP_UTCL1_TRANSLATION_HIT_sum", 0), m.get("TCP_UTCL1_TRANSLATION_MISS_sum", 0),
            m.get("TCP_UTCL1_TRANSLATION_MISS_sum", 0) / req if req else 0))
