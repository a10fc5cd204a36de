"""Timing of the light full-frame cases of the linear kernel on a 4096-frame batch of 480x320: fills and RGB565 maps,
opaque / masked / translucent, with the bytes each has to move."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from vvc_amd import lvgpu  # noqa: E402

lvgpu.init(0)
N, W, H = 4096, 480, 320
area = (0, 0, W - 1, H - 1)


def timeit(fn, K=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K


dst = lvgpu.synth(lvgpu.SYNTH_U16, 1, 0, (N, H, W))
src = lvgpu.synth(lvgpu.SYNTH_U16, 2, 0, (N, H, W))
mask = lvgpu.synth(lvgpu.SYNTH_MASK_UI, 3, 0, (N, H, W))
ctx = lvgpu.DrawCtx(dst)
px = N * W * H
cases = [("opaque fill", lvgpu.BlendDsc(area, color=0x1234, opa=255), 2),
         ("masked fill opa 255", lvgpu.BlendDsc(area, color=0x1234, mask_buf=mask, opa=255), 5),
         ("masked fill opa 200", lvgpu.BlendDsc(area, color=0x1234, mask_buf=mask, opa=200), 5),
         ("translucent fill opa 128", lvgpu.BlendDsc(area, color=0x1234, opa=128), 4),
         ("opaque RGB565 map", lvgpu.BlendDsc(area, src_buf=src, opa=255), 4),
         ("RGB565 map opa 128", lvgpu.BlendDsc(area, src_buf=src, opa=128), 6),
         ("RGB565 map masked opa 255", lvgpu.BlendDsc(area, src_buf=src, mask_buf=mask, opa=255), 7),
         ("RGB565 map masked opa 200", lvgpu.BlendDsc(area, src_buf=src, mask_buf=mask, opa=200), 7)]
for name, dsc, bpp in cases:
    dt = timeit(lambda: lvgpu.lv_draw_sw_blend(ctx, dsc))
    print("%-28s %-16s %7.3f ms %7.1f Gpx/s %6.0f GB/s (%d B/px) %4.1f %%" % (
        name, lvgpu.last_kernel(), dt * 1e3, px / dt / 1e9, px * bpp / dt / 1e9, bpp, px * bpp / dt / 8e10), flush=True)
