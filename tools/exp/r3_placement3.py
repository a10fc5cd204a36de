#!/usr/bin/env python3
"""The fused quarter-turn launch with its three buffers carved out of ONE allocation of most of the device's memory, at
a swept base offset: where in the device's memory do they run fast?  (GPU box)"""
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
GB, MB = 1 << 30, 1 << 20
total = int(sys.argv[1]) if len(sys.argv) > 1 else 264
step = float(sys.argv[2]) if len(sys.argv) > 2 else 8
base = torch.empty(total * GB, dtype=torch.uint8, device="cuda")
a0 = (-base.data_ptr()) % (2 * MB)
sz16, sz32 = px * 2, px * 4


def up(v, a=2 * MB):
    return (v + a - 1) // a * a


def rate(fn, steps=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps * 1e-3


def trial(o_dst, o_src, o_out):
    dst = base[a0 + o_dst: a0 + o_dst + sz16].view(torch.int16).view(n, H, W)
    src = base[a0 + o_src: a0 + o_src + sz32].view(torch.int32).view(n, H, W)
    out = base[a0 + o_out: a0 + o_out + sz16].view(torch.int16).view(n, W, H)
    lvgpu._lib.check(lvgpu._lib.lib().lvgpu_synth(dst.data_ptr(), 0, 1, 0, dst.numel(), None))
    lvgpu._lib.check(lvgpu._lib.lib().lvgpu_synth(src.data_ptr(), 3, 2, 0, src.numel(), None))
    pb = lvgpu.PreparedBlend(lvgpu.DrawCtx(dst), lvgpu.BlendDsc(area, src_buf=src, src_format=lvgpu.SRC_ARGB8888), 1, 0, out)
    return 100 * 10 * px / rate(pb.run) / 8e12


print("dst at 0, src at 2 GB, out swept:", flush=True)
for po in [70 + 2 * i for i in range(20)] + [170 + 2 * i for i in range(20)]:
    print("out at %3d GB: %5.1f" % (po, trial(0, 2 * GB, po * GB)), flush=True)
print("dst at 120, src at 122 GB, out swept:", flush=True)
for po in list(range(0, 118, 8)) + [126 + 4 * i for i in range(32)]:
    print("out at %3d GB: %5.1f" % (po, trial(120 * GB, 122 * GB, po * GB)), flush=True)
