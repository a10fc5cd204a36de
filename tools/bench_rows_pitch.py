import os, sys
sys.path.insert(0, "/root/repo")
import torch
from vvc_amd import lvgpu
lvgpu.init(0)
N = 4096
for (W, H, ba, name) in [(480, 320, (11, 10, 211, 110), "201x101 in 480-wide buffers (pitch 960 B)"),
                         (208, 320, (3, 10, 203, 110), "201x101 in 208-wide buffers (pitch 416 B)"),
                         (256, 320, (3, 10, 203, 110), "201x101 in 256-wide buffers (pitch 512 B)"),
                         (480, 112, (11, 10, 211, 110), "201x101 in 480x112 buffers (frames 105 KB apart)")]:
    dst = lvgpu.synth(lvgpu.SYNTH_U16, 1, 0, (N, H, W))
    ctx = lvgpu.DrawCtx(dst)
    bw, bh = ba[2] - ba[0] + 1, ba[3] - ba[1] + 1
    src = lvgpu.synth(lvgpu.SYNTH_U16, 2, 0, (N, bh, bw))
    mask = lvgpu.synth(lvgpu.SYNTH_MASK_UI, 3, 0, (N, bh, bw))
    p = lvgpu.PreparedBlend(ctx, lvgpu.BlendDsc(ba, src_buf=src, mask_buf=mask, opa=255))
    for _ in range(3):
        p.run()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    torch.cuda.synchronize()
    ev[0].record()
    for _ in range(20):
        p.run()
    ev[1].record()
    torch.cuda.synchronize()
    dt = ev[0].elapsed_time(ev[1]) / 20 * 1e-3
    px = bw * bh * N
    print("%-52s %.3f ms %.0f GB/s %.1f %%" % (name, dt * 1e3, px * 7 / dt / 1e9, px * 7 / dt / 8e10), flush=True)
