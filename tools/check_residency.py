#!/usr/bin/env python3
"""Does the residency cap still pay on this machine?

The full-frame kernels ask for dynamic LDS they never touch so that fewer workgroups share a CU (vvc_amd/csrc/lvgpu.hip:
residency_pad and the *_PAD_LDS constants); the sizes sit on plateaus measured on MI355X under ROCm 7.2.  This runs
every bench workload whose launch is capped with the cap and, through LVGPU_NO_RESIDENCY_CAP, without it, and fails
(exit status 1) when a capped launch is more than 1 % slower than the uncapped one -- the sign that a driver accounts
LDS differently or the memory system behaves differently and the constants want re-measuring.

    python3 tools/check_residency.py           (on a GPU box)
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKLOADS = ["blend_flush", "blend", "rotate", "blend_rgb565a8", "blend_flush_rgb565a8", "fill_masked", "map_opaque", "fill_masked_argb", "fill_translucent"]


def run(workload, capped):
    env = dict(os.environ)
    env.pop("LVGPU_NO_RESIDENCY_CAP", None)
    if not capped:
        env["LVGPU_NO_RESIDENCY_CAP"] = "1"
    best = None
    for _ in range(3):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--no-cpu-baseline",
                              "--no-eighth"], env=env, capture_output=True, timeout=300, check=True).stdout.decode()
        r = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
        ms = r["roofline"]["launch_ms_median"]
        best = ms if best is None else min(best, ms)
    return best


def main():
    bad = []
    print("%-22s %12s %12s %8s" % ("workload", "capped ms", "uncapped ms", "gain"))
    for w in WORKLOADS:
        c, u = run(w, True), run(w, False)
        print("%-22s %12.4f %12.4f %+7.1f%%" % (w, c, u, 100.0 * (u / c - 1.0)), flush=True)
        if c > 1.01 * u:
            bad.append(w)
    if bad:
        print("the residency cap LOSES on: %s -- re-measure the *_PAD_LDS constants or set LVGPU_NO_RESIDENCY_CAP" % ", ".join(bad))
        return 1
    print("the residency cap pays (or is neutral) on every capped workload")
    return 0


if __name__ == "__main__":
    sys.exit(main())
