"""Timing of the widened rows on a 4096-frame batch of 480x320: non-normal blend modes (full-frame RGB565 map,
k_blend_linear) and the ARGB8565 destination (k_blend_argb_dst; masked fill and RGB565 map)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from vvc_amd import lvgpu  # noqa: E402

lvgpu.init(0)
N, W, H = 4096, 480, 320
area = (0, 0, W - 1, H - 1)


def timeit(fn, K=10):
    for _ in range(2):
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
for mode, name in ((0, "normal"), (1, "additive"), (2, "subtractive"), (3, "multiply")):
    for m, bpp in ((None, 6), (mask, 7)):
        dsc = lvgpu.BlendDsc(area, src_buf=src, mask_buf=m, opa=200, blend_mode=mode)
        dt = timeit(lambda: lvgpu.lv_draw_sw_blend(ctx, dsc))
        print("RGB565 map %-11s %-7s opa 200  %-16s %7.3f ms %7.1f Gpx/s %6.0f GB/s (%d B/px) %4.1f %%" % (
            name, "masked" if m is not None else "no mask", lvgpu.last_kernel(), dt * 1e3, px / dt / 1e9, px * bpp / dt / 1e9, bpp,
            px * bpp / dt / 8e10), flush=True)
del dst, ctx
d3 = torch.zeros((N, H, W, 3), dtype=torch.uint8, device="cuda")
d3[..., 2] = 128
ctx3 = lvgpu.DrawCtx(d3, screen_transp=1)
for label, dsc, bpp in (("fill, A8 mask, opa 255", lvgpu.BlendDsc(area, color=0x1234, mask_buf=mask, opa=255), 7),
                        ("RGB565 map, A8 mask, opa 200", lvgpu.BlendDsc(area, src_buf=src, mask_buf=mask, opa=200), 9),
                        ("opaque fill", lvgpu.BlendDsc(area, color=0x1234, opa=255), 3)):
    dt = timeit(lambda: lvgpu.lv_draw_sw_blend(ctx3, dsc))
    print("ARGB8565 dst: %-30s %-16s %7.3f ms %7.1f Gpx/s %6.0f GB/s (%d B/px) %4.1f %%" % (
        label, lvgpu.last_kernel(), dt * 1e3, px / dt / 1e9, px * bpp / dt / 1e9, bpp, px * bpp / dt / 8e10), flush=True)
    bl = lvgpu.BlendList(ctx3, [dsc])
    dt = timeit(lambda: bl.run())
    print("   as a one-item display list %-26s %-16s %7.3f ms %7.1f Gpx/s %6.0f GB/s" % (
        "", lvgpu.last_kernel(), dt * 1e3, px / dt / 1e9, px * bpp / dt / 1e9), flush=True)
# the same destination through the any-rectangle kernel: the two clipped areas of bench.py's clipped_* workloads
for (x1, y1, x2, y2) in ((3, 5, 470, 300), (11, 10, 211, 110)):
    aw, ah = x2 - x1 + 1, y2 - y1 + 1
    m2 = lvgpu.synth(lvgpu.SYNTH_MASK_UI, 5, 0, (N, ah, aw))
    s2 = lvgpu.synth(lvgpu.SYNTH_U16, 6, 0, (N, ah, aw))
    apx = N * aw * ah
    for label, dsc, bpp in (("fill, A8 mask, opa 255", lvgpu.BlendDsc((x1, y1, x2, y2), color=0x1234, mask_buf=m2, opa=255), 7),
                            ("RGB565 map, A8 mask, opa 200", lvgpu.BlendDsc((x1, y1, x2, y2), src_buf=s2, mask_buf=m2, opa=200), 9)):
        dt = timeit(lambda: lvgpu.lv_draw_sw_blend(ctx3, dsc))
        print("ARGB8565 dst, %dx%d area at x = %d: %-28s %-16s %7.3f ms %7.1f Gpx/s %6.0f GB/s (%d B/px) %4.1f %%" % (
            aw, ah, x1, label, lvgpu.last_kernel(), dt * 1e3, apx / dt / 1e9, apx * bpp / dt / 1e9, bpp, apx * bpp / dt / 8e10), flush=True)
