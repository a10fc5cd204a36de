"""Development timing: a clipped, misaligned blend area over a 4096-frame batch (k_blend_rows)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vvc_amd import lvgpu
lvgpu.init(0)
N, W, H = 4096, 480, 320
dst = lvgpu.synth(lvgpu.SYNTH_U16, 1, 0, (N, H, W))
for (ba, name) in [((11, 10, 211, 110), "201x101 at odd x, per-frame src+mask"), ((3, 5, 470, 300), "468x296"), ((100, 100, 131, 131), "32x32")]:
    bw, bh = ba[2] - ba[0] + 1, ba[3] - ba[1] + 1
    src = lvgpu.synth(lvgpu.SYNTH_U16, 2, 0, (N, bh, bw))
    mask = lvgpu.synth(lvgpu.SYNTH_MASK_UI, 3, 0, (N, bh, bw))
    ctx = lvgpu.DrawCtx(dst)
    dsc = lvgpu.BlendDsc(ba, src_buf=src, mask_buf=mask, opa=255)
    for _ in range(3):
        lvgpu.lv_draw_sw_blend(ctx, dsc)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 20
    for _ in range(K):
        lvgpu.lv_draw_sw_blend(ctx, dsc)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    px = bw * bh * N
    print("%-40s %s  %.3f ms  %.1f Gpx/s  %.0f GB/s (7 B/px)" % (name, lvgpu.last_kernel(), dt * 1e3, px / dt / 1e9, px * 7 / dt / 1e9))
