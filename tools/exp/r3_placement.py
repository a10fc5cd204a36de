#!/usr/bin/env python3
"""Does the rate of the fused quarter-turn launch depend on where its three buffers lie relative to each other?  The
draw buffers, the images and the panel images are carved out of ONE allocation at swept offsets (GPU box).
    python3 tools/exp/r3_placement.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from vvc_amd import lvgpu  # noqa: E402

W, H, n = 480, 320, 4096
lvgpu.init(0)
px = W * H * n
MB = 1 << 20
sz16, sz32 = px * 2, px * 4
base = torch.empty(sz16 * 2 + sz32 + 3 * 1024 * MB, dtype=torch.uint8, device="cuda")
a0 = (-base.data_ptr()) % (2 * MB)
area = (0, 0, W - 1, H - 1)


def rate(fn, steps=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 100 * 10 * px / (e0.elapsed_time(e1) / steps * 1e-3) / 8e12


def trial(o_dst, o_src, o_out):
    dst = base[a0 + o_dst: a0 + o_dst + sz16].view(torch.int16).view(n, H, W)
    src = base[a0 + o_src: a0 + o_src + sz32].view(torch.int32).view(n, H, W)
    out = base[a0 + o_out: a0 + o_out + sz16].view(torch.int16).view(n, W, H)
    lvgpu._lib.check(lvgpu._lib.lib().lvgpu_synth(dst.data_ptr(), 0, 1, 0, dst.numel(), None))
    lvgpu._lib.check(lvgpu._lib.lib().lvgpu_synth(src.data_ptr(), 3, 2, 0, src.numel(), None))
    pb = lvgpu.PreparedBlend(lvgpu.DrawCtx(dst), lvgpu.BlendDsc(area, src_buf=src, src_format=lvgpu.SRC_ARGB8888), 1, 0, out)
    return rate(pb.run)


def up(v, a=2 * MB):
    return (v + a - 1) // a * a


# layout: [dst][gap1][src][gap2][out]; the gaps are swept
shifts = [0, 256, 4096, 65536, 1 * MB, 2 * MB, 16 * MB, 128 * MB, 512 * MB, 1024 * MB - 4096]
print("order dst, src, out; gap before src / gap before out -> % of peak", flush=True)
for g1 in shifts:
    row = []
    for g2 in shifts:
        o_dst = 0
        o_src = up(sz16) + g1
        o_out = up(o_src + sz32) + g2
        row.append("%5.1f" % trial(o_dst, o_src, o_out))
    print("g1=%11d: %s" % (g1, " ".join(row)), flush=True)
# other orders at zero gaps
for name, (o_dst, o_src, o_out) in {"src,dst,out": (up(sz32), 0, up(up(sz32) + sz16)), "out,src,dst": (up(up(sz16) + sz32), up(sz16), 0),
                                    "src,out,dst": (up(up(sz32) + sz16), 0, up(sz32))}.items():
    print("order %s: %5.1f" % (name, trial(o_dst, o_src, o_out)), flush=True)
