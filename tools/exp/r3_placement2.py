#!/usr/bin/env python3
"""Does the rate of the fused quarter-turn launch depend on how much device memory is held when its buffers are
allocated (a proxy for where in physical memory they land)?  One process, ballast of 0 to 240 GB (GPU box)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from vvc_amd import lvgpu  # noqa: E402

W, H, n = 480, 320, 4096
lvgpu.init(0)
px = W * H * n
area = (0, 0, W - 1, H - 1)


def rate(fn, steps=30):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 100 * 10 * px / (e0.elapsed_time(e1) / steps * 1e-3) / 8e12


def trial():
    dst = lvgpu.synth(lvgpu.SYNTH_U16, 1, 0, (n, H, W))
    src = lvgpu.synth(lvgpu.SYNTH_ARGB_UI, 2, 0, (n, H, W))
    out = torch.empty((n, W, H), dtype=torch.int16, device="cuda")
    pb = lvgpu.PreparedBlend(lvgpu.DrawCtx(dst), lvgpu.BlendDsc(area, src_buf=src, src_format=lvgpu.SRC_ARGB8888), 1, 0, out)
    r = rate(pb.run)
    del pb, dst, src, out
    torch.cuda.empty_cache()
    return r


GB = 1 << 30
step = int(sys.argv[1]) if len(sys.argv) > 1 else 10
ballast = []
held = 0
while held <= 270:
    print("holding %3d GB: %5.1f %5.1f %%" % (held, trial(), trial()), flush=True)
    ballast.append(torch.empty(step * GB, dtype=torch.uint8, device="cuda"))
    torch.cuda.synchronize()
    held += step
