import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from vvc_amd import lvgpu
lvgpu.init(0)
for (W,H,N) in [(240,135,8192),(135,240,8192),(481,321,2048),(480,320,4096)]:
    d = lvgpu.synth(lvgpu.SYNTH_U16, 1, 0, (N,H,W))
    out = torch.empty((N,W,H), dtype=torch.int16, device='cuda')
    for rot in (1,3):
        for _ in range(3): lvgpu.flush_rotate(d, rot, out=out)
        torch.cuda.synchronize(); t0=time.perf_counter()
        for _ in range(10): lvgpu.flush_rotate(d, rot, out=out)
        torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/10
        print(W,H,rot,lvgpu.last_kernel(),'%.3f ms %.0f GB/s'%(dt*1e3, N*W*H*4/dt/1e9))
