"""Long-running fuzz of the display-list path against the oracle (not collected by pytest): random buffer sizes,
random lists, both swap modes, RGB565 and ARGB8565 destinations, with and without the fused flush.
    python tests/perf/fuzz_list.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import lvref_py as O  # noqa: E402
from test_gpu_list import _apply_oracle, _gpu_items, _rand_item  # noqa: E402
from util import rand_mask, to_dev, to_np  # noqa: E402
from vvc_amd import lvgpu as gpu  # noqa: E402


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    gpu.init(0)
    O.lib()
    t0, n_cases, n_items = time.time(), 0, 0
    while time.time() - t0 < budget:
        swap = int(rng.integers(0, 2))
        argb = rng.random() < 0.3
        W = int(rng.choice([8, 16, 24, 64, 72, 80, 128, 136, 200, 256, 320, 480])) if rng.random() < 0.85 else int(rng.integers(1, 300))
        H = int(rng.choice([1, 7, 8, 33, 64, 65, 96, 128, 130, 240])) if rng.random() < 0.7 else int(rng.integers(1, 200))
        N = int(rng.integers(1, 4))
        area = (0, 0, W - 1, H - 1)
        k = int(rng.integers(1, 90))
        items = []
        while len(items) < k:
            it = _rand_item(rng, W, H, N, bool(rng.integers(0, 2)))
            if argb and it["fmt"] >= 2:
                continue
            if rng.random() < 0.1:  # opaque cover: exercises culling / write-only tiles
                it.update(opa=255, mask=None, mode=0, fmt=it["fmt"] if it["fmt"] < 2 else 0)
                if it["fmt"] == 0:
                    it["src"] = None
            items.append(it)
        if argb:
            dst = rng.integers(0, 256, (N, H, W, 3), dtype=np.uint8)
            dst[..., 2] = rand_mask(rng, (N, H, W))
            exp = dst.copy()
            for it in items:
                for f in range(N):
                    s = None if it["src"] is None else (it["src"] if it["src"].ndim == 2 else it["src"][f])
                    m = None if it["mask"] is None else (it["mask"] if it["mask"].ndim == 2 else it["mask"][f])
                    O.blend(exp[f], area, it["clip"], it["ba"], src=s, color=it["color"], mask=m, mask_res=it["mask_res"],
                            opa=it["opa"], blend_mode=it["mode"], swap=swap, screen_transp=1)
            d = to_dev(dst)
            ctx = gpu.DrawCtx(d, area, area, color_16_swap=swap, screen_transp=1)
            dscs, clips, keep = _gpu_items(gpu, items)
            gpu.blend_list(ctx, dscs, clips)
            got = to_np(d, np.uint8)
            ok = np.array_equal(got, exp)
        else:
            dst = rng.integers(0, 65536, (N, H, W), dtype=np.uint16)
            if rng.random() < 0.5:
                dst[:, : max(1, H // 2)] = 0
            exp = _apply_oracle(O, dst, area, items, swap, W)
            d = to_dev(dst)
            ctx = gpu.DrawCtx(d, area, area, color_16_swap=swap)
            dscs, clips, keep = _gpu_items(gpu, items)
            rot = int(rng.integers(0, 5))
            if rot < 4:
                sb = int(rng.integers(0, 2))
                out = gpu.blend_list_flush(ctx, dscs, rot, swap_bytes=sb, clips=clips)
                o = to_np(out, np.uint16).reshape(N, -1)
                ok = all(np.array_equal(o[f], O.flush_frame(exp[f], W, H, rot, sb)) for f in range(N))
            else:
                gpu.blend_list(ctx, dscs, clips)
                ok = True
            ok = ok and np.array_equal(to_np(d, np.uint16), exp)
        n_cases += 1
        n_items += k
        if not ok:
            print("MISMATCH seed %d case %d: swap %d argb %d %dx%d N %d items %d kernel %s" % (
                seed, n_cases, swap, argb, W, H, N, k, gpu.last_kernel()), flush=True)
            sys.exit(1)
        if n_cases % 50 == 0:
            print("%d cases, %d descriptors, %.0f s" % (n_cases, n_items, time.time() - t0), flush=True)
    print("fuzz ok: %d cases, %d descriptors, seed %d" % (n_cases, n_items, seed))


if __name__ == "__main__":
    main()
