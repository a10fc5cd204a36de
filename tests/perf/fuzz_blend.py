"""Long-running fuzz of the single-descriptor entry points against the oracle (not collected by pytest): random
buffer sizes and alignments, every source format, masks, opacities, blend modes, clip areas, both swap modes,
lvgpu_blend_batch and lvgpu_blend_flush_batch, the latter also without the write-back (LVGPU_BLEND_FLAG_DISCARD_DST);
masked fills and maps also with anti-aliasing off (LVGPU_BLEND_FLAG_NO_AA; the oracle gets the mask already rounded,
which is what the reference's in-place rounding amounts to for the destination).
    python tests/perf/fuzz_blend.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import lvref_py as O  # noqa: E402
from test_gpu_list import _apply_oracle, _rand_item  # noqa: E402
from util import to_dev, to_np  # noqa: E402
from vvc_amd import lvgpu as gpu  # noqa: E402


def argb_case(rng, swap, W, H, N, area, it, kernels):
    """The same item into a 3-byte ARGB8565 draw buffer (screen_transp) placed at a random byte offset inside a
    guarded allocation; an image with its own alpha reaches the blend as a colour plane plus its alpha as the mask
    (lv_draw_sw_img.c:133-136)."""
    dst = rng.integers(0, 256, (N, H, W, 3), dtype=np.uint8)
    dst[..., 2] = rng.choice([0, 255, 255, 1, 2, 3, 128, 252, 253, 254], (N, H, W)).astype(np.uint8) if rng.random() < 0.5 \
        else rng.integers(0, 256, (N, H, W), dtype=np.uint8)
    exp = dst.copy()
    bw, bh = it["ba"][2] - it["ba"][0] + 1, it["ba"][3] - it["ba"][1] + 1
    for f in range(N):
        s = None if it["src"] is None else (it["src"] if it["src"].ndim == (2 if it["fmt"] != 4 else 3) else it["src"][f])
        m = None if it["mask"] is None else (it["mask"] if it["mask"].ndim == 2 else it["mask"][f])
        if it["fmt"] in (0, 1):
            O.blend(exp[f], area, it["clip"], it["ba"], src=s, color=it["color"], mask=m, mask_res=it["mask_res"],
                    opa=it["opa"], blend_mode=it["mode"], swap=swap, screen_transp=1)
        else:
            c16, a8 = O.image_split(np.ascontiguousarray(s), it["fmt"], swap)
            O.blend(exp[f], area, it["clip"], it["ba"], src=c16.reshape(bh, bw), mask=a8.reshape(bh, bw), opa=it["opa"], swap=swap,
                    screen_transp=1)
    off = int(rng.choice([0, 0, 16, 1, 2, 3, 5, 48]))
    n_bytes = N * H * W * 3
    base = torch.zeros(n_bytes + 64, dtype=torch.uint8, device="cuda")
    d = base[off:off + n_bytes].view(N, H, W, 3)
    d.copy_(to_dev(dst))
    ctx = gpu.DrawCtx(d, area, it["clip"], color_16_swap=swap, screen_transp=1)
    sdev = None if it["src"] is None else to_dev(it["src"])
    mdev = None if it["mask"] is None else to_dev(it["mask"])
    gpu.lv_draw_sw_blend(ctx, gpu.BlendDsc(it["ba"], src_buf=sdev, color=it["color"], mask_buf=mdev, mask_res=it["mask_res"],
                                           opa=it["opa"], blend_mode=it["mode"], src_format=it["fmt"]))
    kernels[gpu.last_kernel()] = kernels.get(gpu.last_kernel(), 0) + 1
    ok = np.array_equal(to_np(d.contiguous(), np.uint8), exp) and int(base[:off].sum()) == 0 and int(base[off + n_bytes:].sum()) == 0
    if not ok:
        print("MISMATCH (ARGB8565 destination): swap %d %dx%d N %d off %d item %s kernel %s" % (
            swap, W, H, N, off, {k: v for k, v in it.items() if k not in ("src", "mask")}, gpu.last_kernel()), flush=True)
        sys.exit(1)
    return 1


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    gpu.init(0)
    O.lib()
    t0, n_cases = time.time(), 0
    kernels = {}
    while time.time() - t0 < budget:
        swap = int(rng.integers(0, 2))
        W = int(rng.choice([8, 16, 64, 72, 128, 200, 320, 480])) if rng.random() < 0.7 else int(rng.integers(1, 300))
        H = int(rng.choice([1, 8, 33, 64, 96, 128, 240, 264, 328])) if rng.random() < 0.7 else int(rng.integers(1, 200))
        if rng.random() < 0.08:   # tall, narrow frames: workgroups of 9 to 16 wavefronts, two tiles per wavefront
            H = int(rng.choice([520, 600, 601, 648, 720, 840, 1000, 1021, 1080, 1200, 1272, 1304, 1544]))
            W = int(rng.choice([8, 16, 24, 64, 72])) if rng.random() < 0.7 else int(rng.integers(1, 80))
        N = int(rng.integers(1, 4))
        area = (0, 0, W - 1, H - 1)
        it = _rand_item(rng, W, H, N, bool(rng.integers(0, 2)))
        if rng.random() < 0.4:  # whole-buffer blends take the fast paths
            it["ba"] = it["clip"] = area
            bw, bh = W, H
            n = () if (it["src"] is not None and it["src"].ndim == (3 if it["fmt"] == 4 else 2)) else (N,)
            if it["fmt"] == 1:
                it["src"] = rng.integers(0, 65536, n + (bh, bw), dtype=np.uint16)
            elif it["fmt"] in (2, 3):
                it["src"] = rng.integers(0, 1 << 32, n + (bh, bw), dtype=np.uint32)
            elif it["fmt"] == 4:
                it["src"] = rng.integers(0, 256, n + (bh, bw, 3), dtype=np.uint8)
            if it["mask"] is not None:
                it["mask"] = rng.integers(0, 256, (N, bh, bw), dtype=np.uint8)
        if rng.random() < 0.2:
            n_cases += argb_case(rng, swap, W, H, N, area, it, kernels)
            if n_cases % 200 == 0:
                print("%d cases, %.0f s, kernels %s" % (n_cases, time.time() - t0, kernels), flush=True)
            continue
        dst = rng.integers(0, 65536, (N, H, W), dtype=np.uint16)
        if rng.random() < 0.5:
            dst[:, : max(1, H // 2)] = 0
        flags = 0
        it_o = it
        if it["fmt"] in (0, 1) and rng.random() < 0.2:
            flags |= gpu.FLAG_NO_AA
            if it["mask"] is not None:
                it_o = dict(it, mask=np.where(it["mask"] > 128, 255, 0).astype(np.uint8))
        exp = _apply_oracle(O, dst, area, [it_o], swap, W)
        off = int(rng.choice([0, 0, 0, 1, 3, 4, 8]))  # element offset of the batch inside its allocation
        base = torch.zeros(N * H * W + 16, dtype=torch.int16, device="cuda")
        d = base[off:off + N * H * W].view(N, H, W)
        d.copy_(to_dev(dst))
        ctx = gpu.DrawCtx(d, area, it["clip"], color_16_swap=swap)
        s = None if it["src"] is None else to_dev(it["src"])
        m = None if it["mask"] is None else to_dev(it["mask"])
        dsc = gpu.BlendDsc(it["ba"], src_buf=s, color=it["color"], mask_buf=m, mask_res=it["mask_res"], opa=it["opa"],
                           blend_mode=it["mode"], src_format=it["fmt"])
        rot = int(rng.integers(0, 5))
        ok = True
        if rot < 4 and rng.random() < 0.3:
            flags |= gpu.FLAG_DISCARD_DST
        dsc.flags = flags
        if rot < 4:
            sb = int(rng.integers(0, 2))
            out = gpu.blend_flush(ctx, dsc, rot, swap_bytes=sb)
            o = to_np(out, np.uint16).reshape(N, -1)
            ok = all(np.array_equal(o[f], O.flush_frame(exp[f], W, H, rot, sb)) for f in range(N))
        else:
            gpu.lv_draw_sw_blend(ctx, dsc)
        kernels[gpu.last_kernel()] = kernels.get(gpu.last_kernel(), 0) + 1
        # without the write-back the draw buffer keeps its pixels where blend and flush ran as one pass
        kept = dst if (flags & gpu.FLAG_DISCARD_DST) and "(fused)" in gpu.last_kernel() else exp
        ok = ok and np.array_equal(to_np(d.contiguous(), np.uint16), kept) and int(base[:off].abs().sum()) == 0 \
            and int(base[off + N * H * W:].abs().sum()) == 0
        n_cases += 1
        if not ok:
            print("MISMATCH seed %d case %d: swap %d %dx%d N %d off %d rot %d flags %d item %s kernel %s" % (
                seed, n_cases, swap, W, H, N, off, rot, flags, {k: v for k, v in it.items() if k not in ("src", "mask")},
                gpu.last_kernel()), flush=True)
            sys.exit(1)
        if n_cases % 200 == 0:
            print("%d cases, %.0f s, kernels %s" % (n_cases, time.time() - t0, kernels), flush=True)
    print("fuzz ok: %d cases, seed %d, kernels %s" % (n_cases, seed, kernels))


if __name__ == "__main__":
    main()
