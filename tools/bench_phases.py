#!/usr/bin/env python3
"""The quarter-turn flush fused with an ARGB8888 blend (and alone) over frame sizes on and off the tile grid, for
every setting of the kernels' rows-per-phase switch (LVGPU_TILE_NR / LVGPU_COL_NR / LVGPU_ANY_NR = 8, 4, 2) and, for
the any-size kernel, of the wavefronts per workgroup (LVGPU_ANY_WPB).  One process; the switches are read by the
library at every launch.  Prints one line per (size, mode, setting): kernel, microseconds, % of the 8 TB/s HBM peak.
    python tools/bench_phases.py [quick]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vvc_amd import lvgpu  # noqa: E402

SIZES = [(480, 320, 4096), (3840, 2160, 64), (1920, 1080, 256), (480, 328, 4096), (480, 272, 4096), (1024, 600, 1024),
         (466, 466, 2048), (482, 320, 4096), (172, 320, 8192), (481, 321, 4096), (480, 322, 4096), (240, 135, 16384),
         (135, 240, 16384), (390, 390, 4096), (320, 172, 8192), (240, 240, 8192)]
# frames on the 8-pixel grid whose height is off the 32-pixel grid: both quarter-turn kernels take them
BOTH = [(480, 328, 4096), (480, 272, 4096), (464, 464, 2048), (480, 400, 4096), (800, 480 + 8, 2048), (1024, 600, 1024),
        (1280, 720, 512), (1920, 1080, 256), (3840, 2160, 64)]


def timed(fn, steps=20, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps * 1e3  # us


def main():
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    both = len(sys.argv) > 1 and sys.argv[1] == "both"
    lvgpu.init(0)
    for W, H, n in (BOTH if both else (SIZES[:1] + SIZES[6:8] if quick else SIZES)):
        px = W * H * n
        dst = lvgpu.synth(lvgpu.SYNTH_U16, 1, 0, (n, H, W))
        src = lvgpu.synth(lvgpu.SYNTH_ARGB_UI, 2, 0, (n, H, W))
        out = torch.empty((n, W, H), dtype=torch.int16, device="cuda")
        ctx = lvgpu.DrawCtx(dst)
        area = (0, 0, W - 1, H - 1)
        modes = {
            "fused": (10, lvgpu.PreparedBlend(ctx, lvgpu.BlendDsc(area, src_buf=src, src_format=lvgpu.SRC_ARGB8888), 1, 0, out).run),
            "nowb": (8, lvgpu.PreparedBlend(ctx, lvgpu.BlendDsc(area, src_buf=src, src_format=lvgpu.SRC_ARGB8888,
                                                               flags=lvgpu.FLAG_DISCARD_DST), 1, 0, out).run),
            "rotate": (4, lambda: lvgpu.flush_rotate(dst, 1, out=out)),
        }
        for mode, (bpp, fn) in [(m, v) for m, v in modes.items() for _ in range(2 if both else 1)]:
            if both:  # first pass through k_tile_col, second through k_tile_any
                os.environ["LVGPU_QT_KERNEL"] = "any" if os.environ.get("LVGPU_QT_KERNEL") == "col" else "col"
            fn()
            torch.cuda.synchronize()
            kern = lvgpu.last_kernel()
            is_any = "any" in kern
            var = "LVGPU_ANY_NR" if is_any else ("LVGPU_COL_NR" if "col" in kern else "LVGPU_TILE_NR")
            for nr in (0, 8, 4, 2):  # 0: the launcher's own choice
                os.environ.pop(var, None)
                if nr:
                    os.environ[var] = str(nr)
                wpbs = [0]
                if is_any and (mode != "rotate" or both):
                    wpbs = [0, 3, 4, 5, 6, 8] if not both else [0, 5, 6, 7, 8]
                for wpb in wpbs:
                    if wpb:
                        os.environ["LVGPU_ANY_WPB"] = str(wpb)
                    else:
                        os.environ.pop("LVGPU_ANY_WPB", None)
                    us = timed(fn)
                    print("%4dx%-4d n=%-5d %-6s %-18s NR=%d WPB=%-4s %8.1f us %5.1f %%" %
                          (W, H, n, mode, kern, nr, wpb or "auto", us, 100 * bpp * px / (us * 1e-6) / 8e12), flush=True)
                os.environ.pop("LVGPU_ANY_WPB", None)
            os.environ.pop(var, None)
        os.environ.pop("LVGPU_QT_KERNEL", None)
        del dst, src, out, ctx, modes
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
