"""Worker for test_sharding_gloo: each rank processes its frame range with the CPU oracle standing in for the GPU
(the sharding/aggregation logic is what is under test) and the ranks combine order-independent checksums."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import lvref_py as O  # noqa: E402
from vvc_amd import sharding  # noqa: E402

W, H, TOTAL = 64, 40, 13
px = W * H
rank, world, _, dist = sharding.init("gloo")
lo, hi = sharding.frame_range(rank, world, TOTAL)
n = hi - lo
dst = O.synth(O.SYNTH_U16, 0x5EED0001, lo * px, n * px)          # inputs depend on the GLOBAL frame index only
src = O.synth(O.SYNTH_ARGB_UI, 0x5EED0002, lo * px, n * px)
out = np.empty(n * px, np.uint16)
O.argb_blend_flush_batch(dst, src, out, W, H, n, 1, 1, 1)
cs = (O.checksum(dst, lo * px) + 3 * O.checksum(out, lo * px)) & 0xFFFFFFFFFFFFFFFF
sharding.barrier(dist)
t = sharding.max_over_ranks(dist, [float(rank + 1), 2.0])
total = sharding.sum_over_ranks_u64(dist, cs)
if rank == 0:
    print(json.dumps({"world": world, "checksum": total, "tmax": t, "frames": TOTAL}))
if dist is not None:
    dist.barrier()
    dist.destroy_process_group()
