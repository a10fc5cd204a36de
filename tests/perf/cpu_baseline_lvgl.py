#!/usr/bin/env python3
"""CPU baseline with LVGL PROPER (oracle/_ref: the reference's own sources compiled with the firmware lv_conf.h),
run in the build container.  Frames are streamed through a working set far larger than the caches.  The flush
rotate is the restated lv_port.c loop nest (the reference file itself cannot be compiled here).

    python tests/perf/cpu_baseline_lvgl.py [n_procs]
"""
import ctypes as C
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import lvref_py as O  # noqa: E402

W, H = 480, 320
PX = W * H
AREA = np.array([0, 0, W - 1, H - 1], np.int16)


def worker(args):
    cfg, n_frames, reps = args
    L = O.ref(1)
    L.ref_init(W, H)
    dst = O.synth(O.SYNTH_U16, 1, 0, n_frames * PX).reshape(n_frames, H, W).copy()
    src3 = O.synth(O.SYNTH_MASK_UNI, 2, 0, n_frames * PX * 3).reshape(n_frames, H, W, 3).copy()
    src3[..., 2] = O.synth(O.SYNTH_MASK_UI, 3, 0, n_frames * PX).reshape(n_frames, H, W)
    src16 = O.synth(O.SYNTH_U16, 4, 0, n_frames * PX).reshape(n_frames, H, W)
    out = np.empty(PX, np.uint16)
    a = O._p(AREA)
    full = O.Area(0, 0, W - 1, H - 1)
    cb = O._CB(lambda *x: None)
    t0 = time.perf_counter()
    for _ in range(reps):
        for f in range(n_frames):
            if cfg == 1:      # opaque fill + opaque map (lv_color_fill + row memcpy)
                L.ref_blend(O._p(dst[f]), a, a, a, None, 0x1234, None, 2, None, 255, 0, 0)
                L.ref_blend(O._p(dst[f]), a, a, a, O._p(src16[f]), 0, None, 2, None, 255, 0, 0)
            else:             # image with per-pixel alpha: convert_cb + map_normal in <= hor_res px chunks
                L.ref_img_decoded(O._p(dst[f]), a, a, a, O._p(src3[f]), 5, 255, 0, 0, 0)
                if cfg == 3:
                    O.lib().lvref_flush(C.byref(full), O._p(dst[f]), 1, PX // 10, W, H, C.cast(cb, C.c_void_p), None)
    dt = time.perf_counter() - t0
    return n_frames * reps * PX / dt / 1e6


if __name__ == "__main__":
    procs = int(sys.argv[1]) if len(sys.argv) > 1 else (os.cpu_count() or 1)
    print("CPU:", [l.split(":")[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0],
          "| visible CPUs:", os.cpu_count())
    for cfg, name in [(1, "config 1: opaque fill + opaque map"), (2, "config 2: TRUE_COLOR_ALPHA image over RGB565"),
                      (3, "config 3: config 2 + flush rotate 90 (restated lv_port.c loop)")]:
        n = 256  # 256 frames: 75 MB dst + 113 MB src per process
        one = worker((cfg, n, 2))
        with mp.get_context("fork").Pool(procs) as pool:
            many = sum(pool.map(worker, [(cfg, n, 2)] * procs))
        print("%-62s 1 thread %8.1f Mpx/s | %d processes %9.1f Mpx/s" % (name, one, procs, many))
