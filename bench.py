#!/usr/bin/env python3
"""Benchmark of the batched LVGL pixel path on MI355X.

One "step" = one pass of the hot path over one batch of synthetic frames, device-resident:
    workload "blend_flush" (default): an ARGB8888 image alpha-blended over every RGB565 480x320 draw buffer
        (lv_draw_sw_blend, map_normal with per-pixel alpha) fused with the 90-degree flush rotation into
        the 320x480 panel image -- 4 B + 2 B read, 2 B + 2 B written per pixel = 10 algorithmic bytes/px.
    workload "blend": the blend alone (8 bytes/px).   workload "rotate": the flush rotation alone (4 bytes/px).
    workload "scene": a widget scene recorded from the real LVGL (tests/golden/scene2_*.npz, 1649 blend descriptors)
        replayed over every frame as ONE display list with the fused flush; bytes/px counts what must move (draw
        buffer and panel image written once each, 4 bytes/px) -- this kernel is bound by VALU issue, not by HBM.
Frames are sharded over ranks (one process per GPU), there is no data-path collective; torch.distributed is used
only for the barrier and the max-over-ranks time.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from vvc_amd import lvgpu, sharding  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak
SEED_DST, SEED_SRC, SEED_MASK = 0x5EED0001, 0x5EED0002, 0x5EED0003
BYTES_PER_PX = {"blend_flush": 10, "blend": 8, "rotate": 4, "blend_rgb565a8": 7, "blend_flush_rgb565a8": 9, "scene": 4}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="blend_flush", choices=sorted(BYTES_PER_PX))
    ap.add_argument("--frames-per-gpu", type=int, default=4096)
    ap.add_argument("--width", type=int, default=480, help="logical LVGL width (panel is rotated)")
    ap.add_argument("--height", type=int, default=320)
    ap.add_argument("--rotation", type=int, default=1)
    ap.add_argument("--swap", type=int, default=1, help="LV_COLOR_16_SWAP of the buffers (firmware: 1)")
    ap.add_argument("--mask", default="ui", choices=["ui", "uniform"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-eighth", action="store_true", help="skip the 32-CU (1/8 of the CUs) measurement")
    ap.add_argument("--cpu-threads", type=int, default=0)
    return ap.parse_args()


def cpu_baseline(args, W, H, d0_first_frames_after_one_step, out_first):
    """The CPU oracle (a port of the reference's algorithm: chunked colour/alpha split + map_normal + the flush
    callback's chunked rotate) timed on this host's cores over a bounded sample of the same workload.  Also
    cross-checks the GPU's result for the first frames."""
    import numpy as np
    from oracle import lvref_py as O
    O.build()
    threads = args.cpu_threads or min(16, os.cpu_count() or 1)
    px = W * H
    n = 256
    kind = O.SYNTH_ARGB_UI if args.mask == "ui" else O.SYNTH_ARGB_UNI
    dst = O.synth(O.SYNTH_U16, SEED_DST, 0, n * px)
    src = O.synth(kind, SEED_SRC, 0, n * px)
    out = np.empty(n * px, np.uint16)
    want_out = args.workload != "blend"
    # parity of the first frames (one step applied to fresh inputs)
    chk = dst[: 4 * px].copy()
    chk_out = np.empty(4 * px, np.uint16)
    O.argb_blend_flush_batch(chk, src, chk_out, W, H, 4, args.rotation, args.swap, threads)
    ok = bool(np.array_equal(chk, d0_first_frames_after_one_step))
    if out_first is not None:
        ok = ok and bool(np.array_equal(chk_out, out_first))
    t0 = time.perf_counter()
    O.argb_blend_flush_batch(dst, src, out if want_out else None, W, H, n, args.rotation, args.swap, threads)
    once = time.perf_counter() - t0
    reps = max(1, min(200, int(1.5 / max(once, 1e-4))))
    t0 = time.perf_counter()
    for _ in range(reps):
        O.argb_blend_flush_batch(dst, src, out if want_out else None, W, H, n, args.rotation, args.swap, threads)
    dt = time.perf_counter() - t0
    ref_lvgl = None
    try:  # the reference's own blend (LVGL compiled from its sources into oracle/_ref), one thread, blend part only
        L = O.ref(args.swap)
        if L is not None:
            L.ref_init(W, H)
            nf = 512
            d3 = O.synth(O.SYNTH_U16, SEED_DST, 0, nf * px).reshape(nf, H, W).copy()
            s3 = O.synth(O.SYNTH_MASK_UNI, SEED_SRC, 0, nf * px * 3).reshape(nf, H, W, 3).copy()
            s3[..., 2] = O.synth(O.SYNTH_MASK_UI if args.mask == "ui" else O.SYNTH_MASK_UNI, SEED_MASK, 0, nf * px).reshape(nf, H, W)
            a16 = O._p(np.array([0, 0, W - 1, H - 1], np.int16))
            t0 = time.perf_counter()
            for f in range(nf):
                L.ref_img_decoded(O._p(d3[f]), a16, a16, a16, O._p(s3[f]), 5, 255, 0, 0, 0)
            rdt = time.perf_counter() - t0
            ref_lvgl = {"value": round(nf * px / rdt / 1e6, 1), "unit": "Mpx/s", "cores": 1, "kind": "reference",
                        "sample": "lv_draw_sw_img_decoded (TRUE_COLOR_ALPHA image over RGB565, blend only, no flush), "
                                  "%d frames of %dx%d, %.1f s" % (nf, W, H, rdt)}
    except Exception as e:  # the reference build is optional on the box
        ref_lvgl = {"error": str(e)}
    return {
        "reference_lvgl": ref_lvgl,
        "value": round(n * px * reps / dt / 1e6, 1), "unit": "Mpx/s", "cores": threads, "kind": "port",
        "sample": "%d frames of %dx%d x %d passes, %s, %.1f s wall on %d threads (OpenMP over frames)"
                  % (n, W, H, reps, args.workload, dt, threads),
        "gpu_matches_cpu_on_first_frames": ok,
    }


def cpu_baseline_scene(args, W, H, rec, d_first, o_first):
    """The scene workload on the host: the oracle replays the recorded descriptors frame by frame (OpenMP over frames)
    and applies the restated flush rotation; where the real LVGL (oracle/_ref) is present it also renders the same
    widget scene itself, rasterisers included, on one thread."""
    import ctypes as C
    import numpy as np
    from oracle import lvref_py as O
    O.build()
    threads = args.cpu_threads or min(16, os.cpu_count() or 1)
    px = W * H
    n = 256
    prepared = O.prepare_list([dict(area=r["area"], src=r["src"], mask=r["mask"], color=r["color"], opa=r["opa"],
                                    blend_mode=r["blend_mode"]) for r in rec], W, H, args.swap)
    dst = O.synth(O.SYNTH_U16, SEED_DST, 0, n * px).reshape(n, H, W).copy()
    out = np.empty(n * px, np.uint16)
    O.list_flush_batch(dst, prepared, out, n, args.rotation, threads)
    ok = bool(np.array_equal(dst[:4].reshape(-1), d_first)) and bool(np.array_equal(out[: 4 * px], o_first))
    t0 = time.perf_counter()
    O.list_flush_batch(dst, prepared, out, n, args.rotation, threads)
    once = time.perf_counter() - t0
    reps = max(1, min(1000, int(1.5 / max(once, 1e-4))))
    t0 = time.perf_counter()
    for _ in range(reps):
        O.list_flush_batch(dst, prepared, out, n, args.rotation, threads)
    dt = time.perf_counter() - t0
    ref_lvgl = None
    try:
        L = O.ref(args.swap)
        if L is not None:
            O.render_scene(L, W, H, 2, 0)
            k = 50
            t0 = time.perf_counter()
            for _ in range(k):
                L.ref_scene_build(2)   # invalidates the screen again
                L.ref_scene_refresh()
            rdt = time.perf_counter() - t0
            L.ref_scene_display(16, 16, None, None, 0, None, None, 1, 0)
            ref_lvgl = {"value": round(k * px / rdt / 1e6, 1), "unit": "Mpx/s", "cores": 1, "kind": "reference",
                        "sample": "the real LVGL builds and renders the same widget scene (rasterisers + blend, no flush "
                                  "rotation) %d times, %.2f ms per frame" % (k, rdt / k * 1e3)}
    except Exception as e:
        ref_lvgl = {"error": str(e)}
    return {
        "reference_lvgl": ref_lvgl,
        "value": round(n * px * reps / dt / 1e6, 1), "unit": "Mpx/s", "cores": threads, "kind": "port",
        "sample": "%d frames of %dx%d x %d passes, %d descriptors per frame + flush rotation, %.1f s wall on %d threads "
                  "(OpenMP over frames)" % (n, W, H, reps, len(rec), dt, threads),
        "gpu_matches_cpu_on_first_frames": ok,
    }


def eighth_gpu(args, step_on, n_px, bpp, copy_on=None):
    """The same launch restricted to 32 of the 256 CUs (hipExtStreamCreateWithCUMask), i.e. 1/8 of the GPU's compute.
    The runtime deals CU-mask bits round-robin over the XCDs and rejects a mask that leaves an XCD empty, so the
    32 CUs are 4 per XCD; the memory system behind them is the whole device's."""
    try:
        ms = lvgpu.MaskedStream(range(32))
    except lvgpu.LvgpuError as e:
        return {"error": str(e)}
    census = ms.census()
    for _ in range(2):
        step_on(ms.handle)
    ms.synchronize()
    k = 5
    t0 = time.perf_counter()
    for _ in range(k):
        step_on(ms.handle)
    ms.synchronize()
    dt = (time.perf_counter() - t0) / k
    copy_gbs = None
    if copy_on is not None:  # the empirical ceiling of the same 32 CUs: a plain device copy (2 B read + 2 B written per px)
        copy_on(ms.handle)
        ms.synchronize()
        t0 = time.perf_counter()
        for _ in range(k):
            copy_on(ms.handle)
        ms.synchronize()
        copy_gbs = 4 * n_px / ((time.perf_counter() - t0) / k) / 1e9
    ms.close()
    gbs = bpp * n_px / dt / 1e9
    res = {"cus": 32, "workgroups_per_xcd": census, "value": round(n_px / dt / 1e6, 1), "unit": "Mpx/s",
           "achieved_GBs": round(gbs, 1), "nominal_peak_GBs": HBM_PEAK_GBS / 8,
           "frac_of_nominal": round(gbs / (HBM_PEAK_GBS / 8), 4)}
    if copy_gbs is not None:
        res["measured_copy_GBs"] = round(copy_gbs, 1)
        res["frac_of_measured_copy"] = round(gbs / copy_gbs, 4)
    return res


def main():
    args = parse()
    # LVGPU_BENCH_BACKEND=gloo + LVGPU_BENCH_ONE_DEVICE=1 rehearse the multi-rank code path on a single GPU
    # (all ranks share device 0; the numbers are then meaningless, the control flow is what is exercised).
    backend = os.environ.get("LVGPU_BENCH_BACKEND", "nccl")
    one_dev = os.environ.get("LVGPU_BENCH_ONE_DEVICE") == "1"
    if one_dev:
        os.environ["LOCAL_RANK_REAL"] = os.environ.get("LOCAL_RANK", "0")
        os.environ["LOCAL_RANK"] = "0"
    rank, world, local_rank, dist = sharding.init(backend if int(os.environ.get("WORLD_SIZE", "1")) > 1 else None)
    torch.cuda.set_device(local_rank)
    dev = torch.cuda.current_device()
    lvgpu.init(dev)

    W, H, n = args.width, args.height, args.frames_per_gpu
    px = W * H
    lo, hi = sharding.frame_range(rank, world, n * world)  # this rank's global frame indices
    assert hi - lo == n
    first = lo * px  # global element index of this shard: inputs depend only on the global frame index
    kind = lvgpu.SYNTH_ARGB_UI if args.mask == "ui" else lvgpu.SYNTH_ARGB_UNI
    dst = lvgpu.synth(lvgpu.SYNTH_U16, SEED_DST, first, (n, H, W))
    src = lvgpu.synth(kind, SEED_SRC, first, (n, H, W))
    rot = args.rotation
    out_shape = (n, W, H) if rot in (1, 3) else (n, H, W)
    out = torch.empty(out_shape, dtype=torch.int16, device="cuda")
    ctx = lvgpu.DrawCtx(dst, color_16_swap=args.swap)
    area = (0, 0, W - 1, H - 1)
    dsc = lvgpu.BlendDsc(area, src_buf=src, opa=255, src_format=lvgpu.SRC_ARGB8888)
    if args.workload.endswith("rgb565a8"):  # LV_IMG_CF_RGB565A8: colour plane + A8 plane (lv_draw_sw_img.c:85-95)
        src16 = lvgpu.synth(lvgpu.SYNTH_U16, SEED_SRC, first, (n, H, W))
        a8 = lvgpu.synth(lvgpu.SYNTH_MASK_UI if args.mask == "ui" else lvgpu.SYNTH_MASK_UNI, SEED_MASK, first, (n, H, W))
        dsc = lvgpu.BlendDsc(area, src_buf=src16, mask_buf=a8, opa=255)
    swap_bytes = 0 if args.swap else 1
    blist, rec = None, None
    if args.workload == "scene":
        assert (W, H) == (480, 320), "the recorded scenes are 480x320"
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from util import load_scene
        _fb, rec = load_scene(os.path.join(ROOT, "tests", "golden", "scene2_f0_swap%d.npz" % args.swap))
        dscs, keep = [], []
        for r in rec:
            s16 = None if r["src"] is None else torch.from_numpy(r["src"].view("int16").copy()).cuda()
            m8 = None if r["mask"] is None else torch.from_numpy(r["mask"].copy()).cuda()
            keep += [s16, m8]
            dscs.append(lvgpu.BlendDsc(r["area"], src_buf=s16, color=r["color"], mask_buf=m8, opa=r["opa"],
                                       blend_mode=r["blend_mode"]))
        blist = lvgpu.BlendList(ctx, dscs).compile(flush=True, rotation=rot, swap_bytes=swap_bytes)  # planned once

    def step(stream=None):
        if blist is not None:
            blist.replay(out=out, stream=stream)
        elif args.workload.startswith("blend_flush"):
            lvgpu.blend_flush(ctx, dsc, rot, swap_bytes=swap_bytes, out=out, stream=stream)
        elif args.workload.startswith("blend"):
            lvgpu.lv_draw_sw_blend(ctx, dsc, stream=stream)
        else:
            lvgpu.flush_rotate(dst, rot, swap_bytes=swap_bytes, out=out, stream=stream)

    # first step on fresh inputs: keep the first frames for the CPU cross-check
    step()
    torch.cuda.synchronize()
    kernel = lvgpu.last_kernel()
    d_first = dst[:4].cpu().numpy().view("uint16").reshape(-1).copy()
    o_first = out[:4].cpu().numpy().view("uint16").reshape(-1).copy() if ("flush" in args.workload or args.workload == "scene") else None
    for _ in range(max(0, args.warmup - 1)):
        step()

    def barrier():
        sharding.barrier(dist)

    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)  # on the launch stream
    barrier()
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        step()
    ev1.record()
    barrier()
    dt = time.perf_counter() - t0
    ev_ms = ev0.elapsed_time(ev1)
    # outside the timed region: a few launches bracketed one by one, for the median launch time
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(12)]
    evs[0].record()
    for i in range(11):
        step()
        evs[i + 1].record()
    torch.cuda.synchronize()
    per_step = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(11))
    ev_median = per_step[5]
    dt, ev_ms, ev_median = sharding.max_over_ranks(dist, [dt, ev_ms, ev_median])

    eighth = None
    if rank == 0 and world == 1 and not args.no_eighth:
        eighth = eighth_gpu(args, lambda st: step(st.value), px * n, BYTES_PER_PX[args.workload],
                            copy_on=lambda st: lvgpu.flush_rotate(dst, 0, out=out.view(dst.shape), stream=st.value))
        torch.cuda.synchronize()

    if rank == 0:
        total_px = px * n * world * args.steps
        bpp = BYTES_PER_PX[args.workload]
        launch_s = ev_ms / 1e3 / args.steps  # one kernel launch per step: HIP events on the launch stream
        achieved = bpp * px * n / launch_s / 1e9
        traffic = None
        tj = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tj) and (n, W, H) == (4096, 480, 320):  # measured for exactly this launch shape
            try:
                traffic = json.load(open(tj)).get(args.workload)
            except Exception:
                traffic = None
        res = {
            "metric": "Mpixels/sec batched 320x480 RGB565 blend+rotate; % of HBM peak at 1/8 GPU",
            "value": round(total_px / dt / 1e6, 1),
            "unit": "Mpx/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u16",
            "data": "synthetic",
            "config": {
                "workload": ("scene: %d LVGL blend descriptors (recorded widget scene) as one compiled display list + flush, "
                             "LV_COLOR_16_SWAP %d, %dx%d logical, rotation %d, %d frames per GPU"
                             % (blist.n, args.swap, W, H, rot * 90, n)) if blist is not None else
                            "%s: image with per-pixel alpha over RGB565 (%s alpha), opa 255, LV_COLOR_16_SWAP %d, %dx%d logical, "
                            "rotation %d, %d frames per GPU" % (args.workload, args.mask, args.swap, W, H, rot * 90, n),
                "frames_per_gpu": n, "total_frames": n * world, "parallelism": "frames sharded, no collective",
            },
            "roofline": {
                "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "kernel": kernel, "algorithmic_bytes_per_px": bpp, "launch_ms": round(launch_s * 1e3, 4),
                "launch_ms_median": round(ev_median, 4),
            },
            "pct_hbm_peak": round(100 * achieved / HBM_PEAK_GBS, 2),
        }
        if eighth is not None:
            res["eighth_gpu"] = eighth
        if world == 1 and not args.no_cpu_baseline and args.workload in ("blend_flush", "blend"):
            res["cpu_baseline"] = cpu_baseline(args, W, H, d_first, o_first)
        if world == 1 and not args.no_cpu_baseline and args.workload == "scene":
            res["cpu_baseline"] = cpu_baseline_scene(args, W, H, rec, d_first, o_first)
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
