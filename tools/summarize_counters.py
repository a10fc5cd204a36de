"""Per workload and kernel: the mean per launch of every counter found under <dir>/<workload>/**/*counter_collection.csv
(the layout tools/collect_l2.sh and tools/collect_tlb.sh write).   python tools/summarize_counters.py <dir> [workload ...]"""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
for w in sys.argv[2:] or sorted(os.listdir(root)):
    if not os.path.isdir(os.path.join(root, w)):
        continue
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob("%s/%s/**/*counter_collection.csv" % (root, w), recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in acc.items():
        if "k_synth" in k or "rocclr" in k or "k_first" in k:
            continue
        print("%s  %s" % (w, k))
        for n in sorted(c):
            print("    %-44s %16.0f" % (n, sum(c[n]) / len(c[n])))
