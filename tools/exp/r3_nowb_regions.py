#!/usr/bin/env python3
"""Fused ARGB8888 blend + quarter turn without the write-back (8 B/px): region of the draw buffers, the sources and the
panel images, all 9 assignments with the draw buffers' region called 0.  4096 frames of 480x320; % of 8 TB/s."""
import itertools
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vvc_amd import lvgpu  # noqa: E402
from vvc_amd.placement import RegionArena  # noqa: E402

lvgpu.init(0)
n, W, H = 4096, 480, 320
arena = RegionArena().calibrate(lvgpu)
print(arena.describe())
bufs = {}
for r in (0, 1, 2):
    bufs[r] = (lvgpu.synth(lvgpu.SYNTH_U16, 1, 0, (n, H, W), out=arena.empty((n, H, W), torch.int16, r)),
               lvgpu.synth(lvgpu.SYNTH_ARGB_UI, 2, 0, (n, H, W), out=arena.empty((n, H, W), torch.int32, r)),
               arena.empty((n, W, H), torch.int16, r))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for flags, bpp, name in ((lvgpu.FLAG_DISCARD_DST, 8.0, "without the write-back"), (0, 10.0, "with the write-back")):
    res = []
    for rs, ro in itertools.product((0, 1, 2), repeat=2):
        ctx = lvgpu.DrawCtx(bufs[0][0], color_16_swap=1)
        dsc = lvgpu.BlendDsc((0, 0, W - 1, H - 1), src_buf=bufs[rs][1], opa=255, src_format=lvgpu.SRC_ARGB8888, flags=flags)
        p = lvgpu.PreparedBlend(ctx, dsc, 1, 0, bufs[ro][2])
        p.run()
        e0.record()
        for _ in range(10):
            p.run()
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / 10
        res.append((bpp * n * W * H / (ms * 1e-3) / 8e12 * 100, rs, ro))
    print(name, lvgpu.last_kernel())
    for pct, rs, ro in sorted(res, reverse=True):
        print("  dst 0  src %d  panel %d : %.1f %%" % (rs, ro, pct))
