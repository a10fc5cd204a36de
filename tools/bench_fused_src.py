"""The fused blend + quarter-turn flush per source kind on 4096 frames of 480x320 (HIP events, median of 9 launches).
With LVGPU_NO_BAND=1 in the environment the masked cases stay on the tile kernel: run both ways to compare k_band
with k_tile on the machine at hand.
    python3 tools/bench_fused_src.py"""
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vvc_amd import lvgpu

lvgpu.init(0)
N, W, H = 4096, 480, 320
area = (0, 0, W - 1, H - 1)
dst = lvgpu.synth(lvgpu.SYNTH_U16, 1, 0, (N, H, W))
out = torch.empty((N, W, H), dtype=torch.int16, device="cuda")
src16 = lvgpu.synth(lvgpu.SYNTH_U16, 2, 0, (N, H, W))
a8 = lvgpu.synth(lvgpu.SYNTH_MASK_UI, 3, 0, (N, H, W))
argb = lvgpu.synth(lvgpu.SYNTH_ARGB_UI, 2, 0, (N, H, W))
s8565 = lvgpu.synth(lvgpu.SYNTH_MASK_UI, 4, 0, (N, H, W, 3))
cases = [("ARGB8888 image, opa 255", 10, dict(src_buf=argb, src_format=lvgpu.SRC_ARGB8888, opa=255)),
         ("ARGB8565 image (TRUE_COLOR_ALPHA), opa 255", 9, dict(src_buf=s8565, src_format=lvgpu.SRC_ARGB8565, opa=255)),
         ("ARGB8565 image, opa 200", 9, dict(src_buf=s8565, src_format=lvgpu.SRC_ARGB8565, opa=200)),
         ("RGB565 image + A8 mask, opa 255", 9, dict(src_buf=src16, mask_buf=a8, opa=255)),
         ("RGB565 image + A8 mask, opa 200", 9, dict(src_buf=src16, mask_buf=a8, opa=200)),
         ("fill + A8 mask, opa 255", 7, dict(color=0x1234, mask_buf=a8, opa=255)),
         ("fill + A8 mask, opa 200", 7, dict(color=0x1234, mask_buf=a8, opa=200)),
         ("RGB565 image, opa 200 (no mask)", 8, dict(src_buf=src16, opa=200))]
for name, bpp, kw in cases:
    prep = lvgpu.PreparedBlend(lvgpu.DrawCtx(dst), lvgpu.BlendDsc(area, **kw), 1, 0, out)
    for _ in range(3):
        prep.run(None)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(10)]
    ev[0].record()
    for i in range(9):
        prep.run(None)
        ev[i + 1].record()
    torch.cuda.synchronize()
    ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(9))[4]
    print("%-44s %-18s %.3f ms  %5.1f %% of 8 TB/s" % (name, lvgpu.last_kernel(), ms, bpp * N * W * H / (ms * 1e-3) / 8e12 * 100), flush=True)
