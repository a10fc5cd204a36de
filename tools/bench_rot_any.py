"""The flush alone (4 B/px) on frame sizes off the 8-pixel grid, all four rotations, against the aligned size.
Run on the GPU box: python tools/bench_rot_any.py"""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import torch  # noqa: E402

from vvc_amd import lvgpu  # noqa: E402

lvgpu.init(0)
for (W, H, N) in [(466, 466, 2048), (240, 135, 8192), (135, 240, 8192), (481, 321, 2048), (480, 320, 4096)]:
    d = lvgpu.synth(lvgpu.SYNTH_U16, 1, 0, (N, H, W))
    out = torch.empty((N, W, H), dtype=torch.int16, device='cuda')
    for rot in (0, 1, 2, 3):
        for _ in range(3):
            lvgpu.flush_rotate(d, rot, out=out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            lvgpu.flush_rotate(d, rot, out=out)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        print("%dx%d rot %d %-20s %.3f ms %5.0f GB/s %5.1f %% of 8 TB/s" % (W, H, rot * 90, lvgpu.last_kernel(), dt * 1e3,
                                                                        N * W * H * 4 / dt / 1e9, N * W * H * 4 / dt / 8e10))
