"""Development timing: the fused blend+rotate step when frames start and end in (pinned) host memory, i.e. what a
caller pays who hands over host buffers: H2D of draw buffers + images, kernel, D2H of draw buffers + panel images."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vvc_amd import lvgpu
lvgpu.init(0)
N, W, H = 1024, 480, 320
h_dst = torch.randint(-32768, 32767, (N, H, W), dtype=torch.int16).pin_memory()
h_src = torch.randint(-2**31, 2**31 - 1, (N, H, W), dtype=torch.int32).pin_memory()
h_out = torch.empty((N, W, H), dtype=torch.int16).pin_memory()
h_back = torch.empty((N, H, W), dtype=torch.int16).pin_memory()
d_dst = torch.empty((N, H, W), dtype=torch.int16, device="cuda")
d_src = torch.empty((N, H, W), dtype=torch.int32, device="cuda")
d_out = torch.empty((N, W, H), dtype=torch.int16, device="cuda")
area = (0, 0, W - 1, H - 1)
def step():
    d_dst.copy_(h_dst, non_blocking=True)
    d_src.copy_(h_src, non_blocking=True)
    lvgpu.blend_flush(lvgpu.DrawCtx(d_dst), lvgpu.BlendDsc(area, src_buf=d_src, src_format=lvgpu.SRC_ARGB8888), 1, out=d_out)
    h_back.copy_(d_dst, non_blocking=True)
    h_out.copy_(d_out, non_blocking=True)
for _ in range(2):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 5
for _ in range(K):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
px = N * W * H
print("PCIe-inclusive fused blend+rotate: %.1f Mpx/s (%.1f ms per %d frames; %.1f GB/s over the link, 6 B/px in + 4 B/px out)"
      % (px / dt / 1e6, dt * 1e3, N, px * 10 / dt / 1e9))
