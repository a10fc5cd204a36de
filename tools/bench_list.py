"""Display list vs. one launch per descriptor: a synthetic scene of n_items descriptors (full-screen background, panels,
rows, glyph-sized masked fills, images) replayed over a batch of 480x320 frames and flushed with a 90-degree rotation.
Run on the GPU box:  python tools/bench_list.py [--frames 4096] [--items 300]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vvc_amd import lvgpu  # noqa: E402


def scene(rng, W, H, n_items):
    items = [dict(ba=(0, 0, W - 1, H - 1), opa=255, color=0x2104, mask=None, src=None, fmt=0)]
    while len(items) < n_items:
        r = rng.random()
        if r < 0.05:     # panel
            w, h = int(rng.integers(60, 240)), int(rng.integers(40, 160))
        elif r < 0.55:   # one row of a rounded / anti-aliased shape
            w, h = int(rng.integers(8, 200)), 1
        elif r < 0.9:    # glyph
            w, h = int(rng.integers(4, 14)), int(rng.integers(6, 18))
        else:            # icon
            w, h = 48, 48
        x1, y1 = int(rng.integers(0, W - w)), int(rng.integers(0, H - h))
        it = dict(ba=(x1, y1, x1 + w - 1, y1 + h - 1), opa=int(rng.choice([255, 255, 255, 180, 128])),
                  color=int(rng.integers(0, 65536)), mask=None, src=None, fmt=0)
        if h == 1 or (w < 16 and h < 20):
            it["mask"] = rng.integers(0, 256, (h, w), dtype=np.uint8)
        elif w == 48:
            it["fmt"] = 2
            it["src"] = rng.integers(0, 1 << 32, (h, w), dtype=np.uint32)
        elif it["opa"] != 255:
            it["opa"] = 255 if rng.random() < 0.5 else it["opa"]
        items.append(it)
    return items


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=4096)
    ap.add_argument("--items", type=int, default=300)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--only-list", action="store_true")
    ap.add_argument("--rot", type=int, default=1, help="flush rotation (LV_DISP_ROT_*: 0, 1 = 90, 2 = 180, 3 = 270 degrees)")
    ap.add_argument("--scene", default=None, help="a golden display list recorded from LVGL (tests/golden/scene*.npz)")
    a = ap.parse_args()
    W, H, N = 480, 320, a.frames
    lvgpu.init(0)
    rng = np.random.default_rng(5)
    items = scene(rng, W, H, a.items)
    if a.scene:
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
        from util import load_scene
        fb, rec = load_scene(a.scene)
        items = [dict(ba=r["area"], opa=r["opa"], color=r["color"], mask=r["mask"], fmt=0 if r["src"] is None else 1,
                      src=r["src"], mode=r["blend_mode"]) for r in rec]
    keep, dscs = [], []
    layer_px = 0
    for it in items:
        sv = None if it["src"] is None else np.ascontiguousarray(it["src"]).view(np.int16 if it["fmt"] == 1 else np.int32)
        s = None if sv is None else torch.from_numpy(sv.copy()).cuda()
        m = None if it["mask"] is None else torch.from_numpy(np.ascontiguousarray(it["mask"]).copy()).cuda()
        keep += [s, m]
        dscs.append(lvgpu.BlendDsc(it["ba"], src_buf=s, color=it["color"], mask_buf=m, opa=it["opa"], src_format=it["fmt"],
                                   blend_mode=it.get("mode", 0)))
        layer_px += (it["ba"][2] - it["ba"][0] + 1) * (it["ba"][3] - it["ba"][1] + 1)
    buf = torch.zeros((N, H, W), dtype=torch.int16, device="cuda")
    out = torch.empty((N, W, H), dtype=torch.int16, device="cuda")
    ctx = lvgpu.DrawCtx(buf, color_16_swap=1)

    # descriptors are marshalled once; the timed region holds only the C calls
    import ctypes as C
    from vvc_amd import _lib
    L = _lib.lib()
    arr = lvgpu._fill_list(ctx, dscs, None)
    geom = lvgpu._fill_batch(ctx, lvgpu.BlendDsc(ctx.buf_area))
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    fl = _lib.FlushBatch()
    fl.inp, fl.in_frame_stride, fl.out, fl.out_frame_stride = buf.data_ptr(), H * W, out.data_ptr(), H * W
    fl.width, fl.height, fl.rotation, fl.swap_bytes, fl.n_frames = W, H, a.rot, 0, N

    def run_list():
        _lib.check(L.lvgpu_blend_list_flush_batch(C.byref(geom), arr, len(dscs), out.data_ptr(), H * W, a.rot, 0, st))

    def run_seq():
        for i in range(len(dscs)):
            _lib.check(L.lvgpu_blend_batch(C.byref(arr[i]), st))
        _lib.check(L.lvgpu_flush_rotate_batch(C.byref(fl), st))

    handle = C.c_void_p()
    _lib.check(L.lvgpu_list_create(C.byref(geom), arr, len(dscs), 1, a.rot, 0, C.byref(handle)))

    def run_compiled():
        _lib.check(L.lvgpu_list_replay(handle, None, 0, out.data_ptr(), H * W, st))

    res = {"per_descriptor": float("nan")}
    for name, fn in (("list", run_list), ("compiled_list", run_compiled), ("per_descriptor", run_seq))[: 2 if a.only_list else 3]:
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(a.steps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / a.steps
        dt = e0.elapsed_time(e1) / 1e3 / a.steps
        print("   wall %.3f ms  gpu %.3f ms" % (wall * 1e3, dt * 1e3))
        res[name] = dt
        print("%-15s %8.3f ms/scene  %9.1f Mpx/s of panel output  %9.1f Mpx/s of drawn layers  kernel %s" % (
            name, dt * 1e3, N * W * H / dt / 1e6, N * layer_px / dt / 1e6, lvgpu.last_kernel()), flush=True)
    ref = out.clone()
    run_list()
    torch.cuda.synchronize()
    print("items %d frames %d drawn px per frame %d (%.2f layers)  speedup %.2fx  identical %s" % (
        len(items), N, layer_px, layer_px / (W * H), res["per_descriptor"] / res["list"], bool((ref == out).all())))
    # bytes the list pass must move per frame: draw buffer written (read only where the background does not overwrite),
    # panel image written
    print("list pass: %.1f GB/s of (2 B write dst + 2 B write panel) per px" % (N * W * H * 4 / res["list"] / 1e9))


if __name__ == "__main__":
    main()
