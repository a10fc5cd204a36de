#!/usr/bin/env python3
"""Rewrite the numeric columns (kernel us, Mpx/s, % of peak, traffic) of the results table of the design notes from
profiles/<set>/*_pmc_summary.json:  python3 tools/design_table.py r03"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r03"


def grp(n):
    return "{:,}".format(int(round(n))).replace(",", " ")


rows = {}
for fn in os.listdir(os.path.join(ROOT, "profiles", rnd)):
    if not fn.endswith("_pmc_summary.json"):
        continue
    j = json.load(open(os.path.join(ROOT, "profiles", rnd, fn)))
    if "kernel_trace_avg_ns" not in j:
        continue
    ns, bpl, bpp = j["kernel_trace_avg_ns"], j["algorithmic_bytes_per_launch"], j["algorithmic_bytes_per_px"]
    rows[j["workload"]] = (grp(ns / 1e3), grp(round(bpl / bpp / ns * 1e3, -2)), "%.1f" % (j["kernel_trace_frac_of_8TBs"] * 100),
                           "%.3f" % j["traffic_over_algorithmic"])
path = os.path.join(ROOT, "DESIGN.md")
lines = open(path).read().split("\n")
done = set()
for i, l in enumerate(lines):
    m = re.match(r"\| `(blend_flush --swap 0|[a-z0-9_]+)`", l)
    if not m or l.count("|") != 9:
        continue
    wl = "blend_flush_swap0" if m.group(1).startswith("blend_flush --") else m.group(1)
    if wl not in rows:
        continue
    c = l.split("|")
    us, mpx, pct, tr = rows[wl]
    bold = "**" in c[6]
    c[4], c[5], c[6], c[8] = " %s " % us, " %s " % mpx, (" **%s** " % pct) if bold else (" %s " % pct), " %s " % tr
    lines[i] = "|".join(c)
    done.add(wl)
open(path, "w").write("\n".join(lines))
print("updated:", sorted(done))
print("not in the table:", sorted(set(rows) - done))
