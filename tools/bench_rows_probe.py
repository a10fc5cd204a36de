"""Development timing: how the clipped-area blend (per-frame RGB565 image + A8 mask, 7 B/px) depends on the area's
width and alignment, over a 4096-frame batch of 480x320 buffers."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from vvc_amd import lvgpu  # noqa: E402

lvgpu.init(0)
N, W, H = 4096, 480, 320
dst = lvgpu.synth(lvgpu.SYNTH_U16, 1, 0, (N, H, W))
ctx = lvgpu.DrawCtx(dst)
areas = [((11, 10, 211, 110), "201x101 at x=11"), ((8, 10, 207, 110), "200x101 at x=8 (aligned)"),
         ((0, 10, 479, 110), "480x101 full width"), ((16, 10, 271, 110), "256x101 at x=16"), ((11, 10, 211, 310), "201x301 at x=11"),
         ((64, 10, 127, 110), "64x101 at x=64 (128-B rows, aligned)"), ((3, 5, 470, 300), "468x296 at x=3"), ((0, 0, 479, 319), "full frame")]
for ba, name in areas:
    bw, bh = ba[2] - ba[0] + 1, ba[3] - ba[1] + 1
    src = lvgpu.synth(lvgpu.SYNTH_U16, 2, 0, (N, bh, bw))
    mask = lvgpu.synth(lvgpu.SYNTH_MASK_UI, 3, 0, (N, bh, bw))
    p = lvgpu.PreparedBlend(ctx, lvgpu.BlendDsc(ba, src_buf=src, mask_buf=mask, opa=255))
    for _ in range(3):
        p.run()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    torch.cuda.synchronize()
    K = 20
    ev[0].record()
    for _ in range(K):
        p.run()
    ev[1].record()
    torch.cuda.synchronize()
    dt = ev[0].elapsed_time(ev[1]) / K * 1e-3
    px = bw * bh * N
    print("%-40s %-16s %.3f ms  %.1f Gpx/s  %.0f GB/s (7 B/px)  %.1f %%" % (name, lvgpu.last_kernel(), dt * 1e3, px / dt / 1e9, px * 7 / dt / 1e9, px * 7 / dt / 8e10), flush=True)
    del src, mask, p
