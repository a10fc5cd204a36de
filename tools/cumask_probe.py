"""Development probe: which XCDs does a CU mask select?  One mask per process."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vvc_amd import lvgpu
lvgpu.init(0)
which = sys.argv[1]
bits = {"lo32": range(32), "stride8": range(0, 256, 8), "stride8+3": range(3, 256, 8), "all": range(256),
        "lo64": range(64), "stride8x2": list(range(0, 256, 8)) + list(range(1, 256, 8))}[which]
s = lvgpu.MaskedStream(bits)
print(which, s.census(), flush=True)
