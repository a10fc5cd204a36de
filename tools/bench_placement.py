"""Does the fused blend + flush depend on where its three buffers lie relative to each other?  The draw buffers, the
panel images and the ARGB8888 sources are carved out of one allocation at chosen distances (beyond their sizes) and
the launch is timed with HIP events (median of 9).     python3 tools/bench_placement.py [frames]"""
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vvc_amd import lvgpu

lvgpu.init(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
W, H = 480, 320
area = (0, 0, W - 1, H - 1)
sz16, sz32 = N * W * H * 2, N * W * H * 4
pool = torch.empty(sz16 * 2 + sz32 + (1 << 30), dtype=torch.int8, device="cuda")
print("pool at 0x%x" % pool.data_ptr())
ref_dst = lvgpu.synth(lvgpu.SYNTH_U16, 1, 0, (N, H, W))
ref_src = lvgpu.synth(lvgpu.SYNTH_ARGB_UI, 2, 0, (N, H, W))
KB, MB = 1 << 10, 1 << 20
gaps = [0, 256, 4 * KB, 64 * KB, 1 * MB, 2 * MB + 64 * KB, 3 * MB + 4 * KB + 256, 33 * MB + 512 * KB, 150 * MB, 150 * MB + 300 * KB]
for g1 in gaps:
    row = []
    for g2 in gaps:
        o_out = sz16 + g1
        o_src = o_out + sz16 + g2
        dst = pool[0:sz16].view(torch.int16).view(N, H, W)
        out = pool[o_out:o_out + sz16].view(torch.int16).view(N, W, H)
        src = pool[o_src:o_src + sz32].view(torch.int32).view(N, H, W)
        dst.copy_(ref_dst)
        src.copy_(ref_src)
        prep = lvgpu.PreparedBlend(lvgpu.DrawCtx(dst), lvgpu.BlendDsc(area, src_buf=src, src_format=lvgpu.SRC_ARGB8888, opa=255), 1, 0, out)
        for _ in range(3):
            prep.run(None)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(10)]
        ev[0].record()
        for i in range(9):
            prep.run(None)
            ev[i + 1].record()
        torch.cuda.synchronize()
        ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(9))[4]
        row.append(10 * N * W * H / (ms * 1e-3) / 8e12 * 100)
    print("gap dst-out %10d | " % g1 + " ".join("%5.1f" % v for v in row), flush=True)
print("columns: gap out-src = " + " ".join(str(g) for g in gaps))
