#!/usr/bin/env python3
"""Does the rate of the fused quarter-turn launch depend on WHEN in a process it is measured (clocks settling) or on
WHERE its buffers were allocated (what the allocator handed out after earlier allocations were freed)?  Measures the
same launch in a fresh process, over ten seconds, then again on buffers allocated after a few gigabytes were allocated
and freed, and with the three buffers carved out of one 2-MB-aligned allocation.
    python tools/probe_drift.py [W H N]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vvc_amd import lvgpu  # noqa: E402


def rate(fn, px, bpp=10, steps=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 100 * bpp * px / (e0.elapsed_time(e1) / steps * 1e-3) / 8e12


def make(W, H, n, carve=False):
    if carve:
        total = n * H * W * 2 * 4 + 3 * (2 << 20)
        base = torch.empty(total, dtype=torch.uint8, device="cuda")
        p = (-base.data_ptr()) % (2 << 20)
        def take(nbytes, dtype, shape):
            nonlocal p
            t = base[p:p + nbytes].view(dtype).view(shape)
            p += (nbytes + (2 << 20) - 1) // (2 << 20) * (2 << 20)
            return t
        dst = take(n * H * W * 2, torch.int16, (n, H, W))
        src = take(n * H * W * 4, torch.int32, (n, H, W))
        out = take(n * H * W * 2, torch.int16, (n, W, H))
        lvgpu._lib.check(lvgpu._lib.lib().lvgpu_synth(dst.data_ptr(), 0, 1, 0, dst.numel(), None))
        lvgpu._lib.check(lvgpu._lib.lib().lvgpu_synth(src.data_ptr(), 3, 2, 0, src.numel(), None))
        keep = base
    else:
        dst = lvgpu.synth(lvgpu.SYNTH_U16, 1, 0, (n, H, W))
        src = lvgpu.synth(lvgpu.SYNTH_ARGB_UI, 2, 0, (n, H, W))
        out = torch.empty((n, W, H), dtype=torch.int16, device="cuda")
        keep = None
    area = (0, 0, W - 1, H - 1)
    pb = lvgpu.PreparedBlend(lvgpu.DrawCtx(dst), lvgpu.BlendDsc(area, src_buf=src, src_format=lvgpu.SRC_ARGB8888), 1, 0, out)
    return pb, (dst, src, out, keep)


def main():
    W, H, n = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (480, 328, 4096)
    lvgpu.init(0)
    px = W * H * n
    t0 = time.time()
    pb, bufs = make(W, H, n)
    print("fresh process, pointers mod 2 MB:", [hex(b.data_ptr() % (2 << 20)) for b in bufs[:3]], flush=True)
    while time.time() - t0 < 8:
        print("  t=%4.1f s  %5.1f %%  %s" % (time.time() - t0, rate(pb.run, px), lvgpu.last_kernel()), flush=True)
        time.sleep(0.4)
    del pb, bufs
    torch.cuda.empty_cache()
    for k in range(3):  # churn: a few other shapes come and go, as in a sweep
        junk = [torch.empty((1 << 30) + k * 12345678, dtype=torch.uint8, device="cuda") for _ in range(5)]
        for j in junk:
            j.fill_(1)
        torch.cuda.synchronize()
        del junk
        torch.cuda.empty_cache()
    pb, bufs = make(W, H, n)
    print("after churn, pointers mod 2 MB:", [hex(b.data_ptr() % (2 << 20)) for b in bufs[:3]])
    for _ in range(4):
        print("  %5.1f %%" % rate(pb.run, px), flush=True)
    del pb, bufs
    torch.cuda.empty_cache()
    pb, bufs = make(W, H, n, carve=True)
    print("one 2-MB-aligned allocation, pointers mod 2 MB:", [hex(b.data_ptr() % (2 << 20)) for b in bufs[:3]])
    for _ in range(4):
        print("  %5.1f %%" % rate(pb.run, px), flush=True)
    # idle, then again: do the clocks come back?
    time.sleep(5)
    print("after 5 s idle: %5.1f %%" % rate(pb.run, px))
    for _ in range(3):
        print("  %5.1f %%" % rate(pb.run, px), flush=True)


if __name__ == "__main__":
    main()
