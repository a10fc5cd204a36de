#!/usr/bin/env python3
"""Turn gpurun_out/prof (see tools/collect_profiles.sh) into the committed summaries under profiles/<round>/ and
profiles/traffic.json (HBM bytes per launch of the dominant kernel, which bench.py reports as roofline.traffic).

FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE counts a wide coalesced streaming read at half its size
(128-byte requests tallied as 64 bytes), so reads are doubled; WRITE_SIZE is exact for 16-byte-per-lane stores."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(ROOT, "gpurun_out", "prof")
dst = os.path.join(ROOT, "profiles", rnd)
os.makedirs(dst, exist_ok=True)
KERNEL = {"blend_flush": "k_tile", "blend": "k_blend_linear", "rotate": "k_tile"}
BPP = {"blend_flush": 10, "blend": 8, "rotate": 4}
traffic = {}


def newest(pattern):
    """gpurun merges every run's files into the same directories (named by process id): take the latest."""
    files = sorted(glob.glob(pattern), key=os.path.getmtime)
    return files[-1:]


for w, kname in KERNEL.items():
    d = os.path.join(src, w)
    if not os.path.isdir(d):
        continue
    stats = newest(os.path.join(d, "kt", "*", "*_kernel_stats.csv"))
    if stats:
        rows = list(csv.reader(open(stats[0])))
        with open(os.path.join(dst, "%s_kernel_stats.csv" % w), "w") as f:
            csv.writer(f).writerows(rows)
    vals = {}
    for c in ("fetch", "write"):
        files = newest(os.path.join(d, c, "*", "*_counter_collection.csv"))
        if not files:
            continue
        v = [float(r["Counter_Value"]) for r in csv.DictReader(open(files[0])) if kname in r["Kernel_Name"]]
        if v:
            vals[c] = sum(v) / len(v)
    if len(vals) == 2:
        rd, wr = vals["fetch"] * 1024 * 2, vals["write"] * 1024
        px = 480 * 320 * 4096
        summ = {"workload": w, "kernel": kname, "FETCH_SIZE_KiB_avg_per_launch": vals["fetch"],
                "WRITE_SIZE_KiB_avg_per_launch": vals["write"], "hbm_read_bytes_per_launch(FETCH_SIZE*1024*2)": rd,
                "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr,
                "algorithmic_bytes_per_launch": BPP[w] * px, "traffic_over_algorithmic": (rd + wr) / (BPP[w] * px)}
        json.dump(summ, open(os.path.join(dst, "%s_pmc_summary.json" % w), "w"), indent=1)
        traffic[w] = rd + wr
        print(w, "traffic/algorithmic = %.3f" % summ["traffic_over_algorithmic"])
if traffic:
    json.dump(traffic, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)

# display-list kernel (tools/collect_list_profile.sh): kernel stats + HBM traffic + issue counters in one summary
lsrc = os.path.join(src, "list")
if os.path.isdir(lsrc):
    import collections
    st = newest(os.path.join(lsrc, "kt", "*", "*_kernel_stats.csv"))
    if st:
        rows = list(csv.reader(open(st[0])))
        with open(os.path.join(dst, "list_scene2_kernel_stats.csv"), "w") as f:
            csv.writer(f).writerows(rows)
    out = {"workload": "display list: golden LVGL scene 2 (1649 descriptors, 432119 drawn px per frame) replayed over 4096 "
                       "frames of 480x320 + fused 90-degree flush", "kernel": "k_blend_list"}
    for c in ("fetch", "write", "sq"):
        files = newest(os.path.join(lsrc, c, "*", "*_counter_collection.csv"))
        if not files:
            continue
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(files[0])):
            if "k_blend_list" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            out[k + "_avg_per_launch"] = sum(v) / len(v)
    if "FETCH_SIZE_avg_per_launch" in out and "WRITE_SIZE_avg_per_launch" in out:
        out["hbm_read_bytes_per_launch(FETCH_SIZE*1024*2)"] = out["FETCH_SIZE_avg_per_launch"] * 2048
        out["hbm_write_bytes_per_launch"] = out["WRITE_SIZE_avg_per_launch"] * 1024
        out["minimum_bytes_per_launch(2 B dst write + 2 B panel write per px)"] = 4 * 480 * 320 * 4096
    json.dump(out, open(os.path.join(dst, "list_scene2_pmc_summary.json"), "w"), indent=1)
    print("list: VALU wave-instructions per launch %.0f" % out.get("SQ_INSTS_VALU_avg_per_launch", float("nan")))
