"""Plan statistics of the recorded scene (GPU box: the descriptors point into device memory)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from util import load_scene
from vvc_amd import lvgpu
lvgpu.init(0)
fb, rec = load_scene(os.path.join(ROOT, "tests/golden/scene2_f0_swap1.npz"))
dscs, keep = [], []
for r in rec:
    s16 = None if r["src"] is None else torch.from_numpy(r["src"].view("int16").copy()).cuda()
    m8 = None if r["mask"] is None else torch.from_numpy(r["mask"].copy()).cuda()
    keep += [s16, m8]
    dscs.append(lvgpu.BlendDsc(r["area"], src_buf=s16, color=r["color"], mask_buf=m8, opa=r["opa"], blend_mode=r["blend_mode"]))
dst = torch.zeros((1, 320, 480), dtype=torch.int16, device="cuda")
bl = lvgpu.BlendList(lvgpu.DrawCtx(dst, color_16_swap=1), dscs)
p = bl.plan(flush=True, rotation=1)
for f in p:
    print(f, p[f])
