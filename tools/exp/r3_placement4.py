#!/usr/bin/env python3
"""Three separate allocations pushed into different memory regions by ballast allocations that are freed again: does
the placement (and the rate) survive the release of the ballast?  (GPU box)"""
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
GB = 1 << 30


def rate(fn, steps=30):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 100 * 10 * px / (e0.elapsed_time(e1) / steps * 1e-3) / 8e12


def run(dst, src, out):
    pb = lvgpu.PreparedBlend(lvgpu.DrawCtx(dst), lvgpu.BlendDsc(area, src_buf=src, src_format=lvgpu.SRC_ARGB8888), 1, 0, out)
    return rate(pb.run)


dst = lvgpu.synth(lvgpu.SYNTH_U16, 1, 0, (n, H, W))
b1 = torch.empty(100 * GB, dtype=torch.uint8, device="cuda")
src = lvgpu.synth(lvgpu.SYNTH_ARGB_UI, 2, 0, (n, H, W))
b2 = torch.empty(92 * GB, dtype=torch.uint8, device="cuda")
out = torch.empty((n, W, H), dtype=torch.int16, device="cuda")
print("pointers (GB from dst): src %.1f out %.1f" % ((src.data_ptr() - dst.data_ptr()) / GB, (out.data_ptr() - dst.data_ptr()) / GB))
print("with the ballast held: %5.1f %%" % run(dst, src, out), flush=True)
del b1, b2
torch.cuda.empty_cache()
print("ballast released:      %5.1f %%" % run(dst, src, out), flush=True)
junk = [torch.empty(20 * GB, dtype=torch.uint8, device="cuda").fill_(3) for _ in range(8)]
torch.cuda.synchronize()
print("160 GB of other data allocated and written since: %5.1f %%" % run(dst, src, out), flush=True)
