"""Development timing of the stand-alone colour entry points over 4096 frames of 480x320 (629 Mpx): lv_color_fill,
lv_color_hex / lv_color_to32 conversions, the image colour/alpha split, lv_draw_sw_buffer_copy."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from vvc_amd import lvgpu  # noqa: E402

lvgpu.init(0)
N, H, W = 4096, 320, 480
n = N * H * W


def timeit(fn, k=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(k):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k * 1e-3


buf = torch.zeros(n, dtype=torch.int16, device="cuda")
xrgb = lvgpu.synth(lvgpu.SYNTH_ARGB_UNI, 1, 0, (n,))
c16 = lvgpu.synth(lvgpu.SYNTH_U16, 2, 0, (N, H, W))
dst = torch.zeros((N, H, W), dtype=torch.int16, device="cuda")
cases = [("lv_color_fill", lambda: lvgpu.color_fill(buf, 0x1234), 2),
         ("lv_color_hex (0x00RRGGBB -> lv_color16_t)", lambda: lvgpu.color_from_rgb888(xrgb, 1), 6),
         ("lv_color_to32", lambda: lvgpu.color_to32(c16, 1), 6),
         ("image split ARGB8888 -> colour + alpha planes", lambda: lvgpu.image_split(xrgb, lvgpu.SRC_ARGB8888, 1), 7),
         ("image convert + recolor", lambda: lvgpu.image_convert(xrgb, lvgpu.SRC_ARGB8888, 1, recolor=0x1234, recolor_opa=100), 7),
         ("lv_draw_sw_buffer_copy, whole frames", lambda: lvgpu.buffer_copy(dst, (0, 0, W - 1, H - 1), c16, (0, 0, W - 1, H - 1)), 4),
         ("lv_draw_sw_buffer_copy, 201x101 at (11,10)", None, 4)]
for name, fn, bpp in cases:
    px = n
    if fn is None:
        px = N * 201 * 101
        fn = lambda: lvgpu.buffer_copy(dst, (11, 10, 211, 110), c16, (11, 10, 211, 110))  # noqa: E731
    dt = timeit(fn)
    print("%-50s %.3f ms %6.0f GB/s (%d B/px) %5.1f %%" % (name, dt * 1e3, px * bpp / dt / 1e9, bpp, px * bpp / dt / 8e10), flush=True)
