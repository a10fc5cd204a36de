#!/usr/bin/env python3
"""Fused RGB565 + A8 blend + quarter turn (k_band): which of its four streams should share a region of the device memory.
4096 frames of 480x320; % of 8 TB/s at 9 B/px per assignment (draw buffers, colour plane, A8 plane, panel images)."""
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
for r in (0, 1, 2):   # one of each kind in every region
    bufs[r] = (lvgpu.synth(lvgpu.SYNTH_U16, 1, 0, (n, H, W), out=arena.empty((n, H, W), torch.int16, r)),
               lvgpu.synth(lvgpu.SYNTH_U16, 2, 0, (n, H, W), out=arena.empty((n, H, W), torch.int16, r)),
               lvgpu.synth(lvgpu.SYNTH_MASK_UI, 3, 0, (n, H, W), out=arena.empty((n, H, W), torch.uint8, r)),
               arena.empty((n, W, H), torch.int16, r))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
res = []
for rd, rs, rm, ro in itertools.product((0, 1, 2), repeat=4):
    if rd != 0:
        continue   # by symmetry: the draw buffers' region is called 0
    ctx = lvgpu.DrawCtx(bufs[rd][0], color_16_swap=1)
    dsc = lvgpu.BlendDsc((0, 0, W - 1, H - 1), src_buf=bufs[rs][1], mask_buf=bufs[rm][2], opa=255)
    p = lvgpu.PreparedBlend(ctx, dsc, 1, 0, bufs[ro][3])
    p.run()
    e0.record()
    for _ in range(10):
        p.run()
    e1.record()
    e1.synchronize()
    ms = e0.elapsed_time(e1) / 10
    res.append((9.0 * n * W * H / (ms * 1e-3) / 8e12 * 100, rd, rs, rm, ro))
print(lvgpu.last_kernel())
for pct, rd, rs, rm, ro in sorted(res, reverse=True):
    print("dst %d  colour %d  a8 %d  panel %d : %.1f %%" % (rd, rs, rm, ro, pct))
