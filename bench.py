#!/usr/bin/env python3
"""Benchmark of the batched LVGL pixel path on MI355X.

One "step" = one pass of the hot path over one batch of synthetic frames, device-resident.  Workloads:
    blend_flush (default)  an ARGB8888 image alpha-blended over every RGB565 480x320 draw buffer (lv_draw_sw_blend,
                           map_normal with per-pixel alpha) fused with the 90-degree flush rotation into the 320x480
                           panel image: 4 B + 2 B read, 2 B + 2 B written per pixel = 10 algorithmic bytes/px.
                           With --swap 0 the buffers are LV_COLOR_16_SWAP 0 and the flush byte-swaps into wire order.
    blend_flush_nowb       blend_flush for a display under full_refresh (the reference's configuration, src/lv_port.c:241:
                           LVGL redraws the whole buffer on every refresh, so the blended draw buffer is dead once it has
                           been flushed): LVGPU_BLEND_FLAG_DISCARD_DST skips the write-back of the draw buffer, 4 B + 2 B
                           read, 2 B written per pixel = 8 algorithmic bytes/px.  Reported beside the default, never instead.
    blend / rotate         the blend alone (8 B/px) / the flush rotation alone (4 B/px)
    blend_rgb565a8, blend_flush_rgb565a8   the same with a colour plane + A8 plane source (7 / 9 B/px)
    blend_flush_4k         blend_flush on 3840x2160 canvases (64 frames per GPU by default)
    blend_flush_466        blend_flush on 466x466 frames (a round AMOLED panel's size: off the 8-pixel grid, every frame
                           at another 4-byte alignment), 2048 frames per GPU by default
    flush_180_466          the flush alone, half turn, on the same frames (pixel count not a multiple of 8)
    clipped_small, clipped_large   lv_draw_sw_blend of a per-frame RGB565 image + A8 mask over a clipped,
                           misaligned area (201x101 at an odd x / 468x296 at x = 3) of 480x320 buffers; pixels
                           counted are the area's (7 B/px)
    clipped_small_argb, clipped_large_argb   the same into a 3-byte ARGB8565 draw buffer (screen_transp):
                           3 R + 3 W + 2 R + 1 R = 9 B/px
    fill_translucent       fill_normal without mask, translucent (premultiplied formula + scan-order pre-pass), 4 B/px
    fill_opaque / map_opaque   lv_color_fill per row (2 B/px, write only) / row copy (4 B/px)
    fill_masked            fill_normal with an A8 mask at opa 255 (5 B/px)
    fill_masked_argb       the same into a 3-byte ARGB8565 draw buffer (screen_transp): 3 R + 3 W + 1 R = 7 B/px
    scene                  a widget scene recorded from the real LVGL (tests/golden/scene2_*.npz, 1649 blend
                           descriptors) replayed over every frame as ONE display list with the fused flush; bytes/px
                           counts what must move (draw buffer and panel image written once each, 4 B/px) -- this
                           kernel is bound by VALU issue, not by HBM.
Buffers are placed by region of the device memory (--placement spread, the default: draw buffers, sources + masks and
panel images in different 96-GB regions of one arena whose region map is found by timing first, vvc_amd/placement.py;
--placement compact leaves them where the allocator puts them; the full default run times the headline launch in
both and reports the second as roofline.compact_placement_frac).
Frames are sharded over ranks (one process per GPU), there is no data-path collective; torch.distributed is used only
for the barrier, the max-over-ranks time and the gather of the per-rank times -- over gloo on host tensors by default
(a path without a collective should not depend on RCCL coming up on the node; LVGPU_BENCH_BACKEND=nccl selects RCCL).
With more than one rank `value` comes from the slowest rank's HIP-event time around its timed steps (the ranks start
together after a barrier + device synchronisation; host time across the closing barrier, which adds the barrier's own
latency, is reported beside it as host_ms_per_step), and `efficiency_vs_1gpu` = value / (N x the one-GPU rate) is printed
when that rate is known: from --ref-1gpu-mpx, or from the line a one-GPU run of the same workload left in
gpurun_out/bench_1gpu_cache.json on this machine within the last six hours.

`--gpus N` with N > 1 and no torchrun environment: this process starts N ranks as a child
(`python -m torch.distributed.run --nproc-per-node N bench.py ...`) before it touches any GPU, relays the child's
output and exit code.  With more than one rank the default batch is 65 536 frames in total (sharded per frame,
"strong" scaling); `--frames-per-gpu` fixes the per-GPU batch instead ("weak").

Prints ONE JSON line on rank 0.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak
SEED_DST, SEED_SRC, SEED_MASK = 0x5EED0001, 0x5EED0002, 0x5EED0003
MULTI_GPU_TOTAL_FRAMES = 65536  # the batch that is sharded over 2/4/8 GPUs
# workload -> (algorithmic bytes per pixel, default logical width, height, default frames per GPU on one GPU)
WORKLOADS = {
    "blend_flush": (10, 480, 320, 4096), "blend_flush_nowb": (8, 480, 320, 4096), "blend": (8, 480, 320, 4096),
    "rotate": (4, 480, 320, 4096),
    "blend_rgb565a8": (7, 480, 320, 4096), "blend_flush_rgb565a8": (9, 480, 320, 4096), "scene": (4, 480, 320, 4096),
    "blend_flush_4k": (10, 3840, 2160, 64), "blend_flush_466": (10, 466, 466, 2048), "flush_180_466": (4, 466, 466, 2048),
    "clipped_small": (7, 480, 320, 4096), "clipped_large": (7, 480, 320, 4096),
    "clipped_small_argb": (9, 480, 320, 4096), "clipped_large_argb": (9, 480, 320, 4096),
    "fill_translucent": (4, 480, 320, 4096), "fill_opaque": (2, 480, 320, 4096), "map_opaque": (4, 480, 320, 4096),
    "fill_masked": (5, 480, 320, 4096), "fill_masked_argb": (7, 480, 320, 4096),
}
CLIPPED_AREAS = {"clipped_small": (11, 10, 211, 110), "clipped_large": (3, 5, 470, 300),
                 "clipped_small_argb": (11, 10, 211, 110), "clipped_large_argb": (3, 5, 470, 300)}
KERNEL_SOURCES = ("lvgpu.hip", "lvgpu_tiles.hip", "lvgpu_tiles_col.hip", "lvgpu_tiles_any.hip", "lvgpu_rows.hip", "lvgpu_launch.hpp", "lvgpu_kernels.hpp", "lvgpu_list.hpp",
                  "lvgpu_px.hpp")


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="blend_flush", choices=sorted(WORKLOADS))
    ap.add_argument("--frames-per-gpu", type=int, default=0,
                    help="frames each GPU processes (default: 4096 on one GPU -- 64 for blend_flush_4k --, "
                         "65536 / N on N > 1 GPUs)")
    ap.add_argument("--width", type=int, default=0, help="logical LVGL width (panel is rotated)")
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--rotation", type=int, default=1)
    ap.add_argument("--swap", type=int, default=1, help="LV_COLOR_16_SWAP of the buffers (firmware: 1)")
    ap.add_argument("--mask", default="ui", choices=["ui", "uniform"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-eighth", action="store_true", help="skip the 32-CU (1/8 of the CUs) measurement")
    ap.add_argument("--region-map", default="auto",
                    help="with --placement spread: 'auto' finds the memory regions inside the arena by timing (0.3 s of "
                         "probe launches of the fused tile kernel); a map printed by `python3 -m vvc_amd.placement` on "
                         "this device skips that, so that a kernel trace holds the workload's launches only")
    ap.add_argument("--placement", default="spread", choices=["spread", "compact"],
                    help="spread: draw buffers, sources and panel images in different thirds of the device memory "
                         "(vvc_amd/placement.py); compact: wherever the allocator puts them, one after the other")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--ref-1gpu-mpx", type=float, default=0.0,
                    help="Mpx/s of the same workload on one GPU: with --gpus N > 1 the line then carries efficiency_vs_1gpu")
    return ap.parse_args(argv)


CACHE_1GPU = os.path.join(ROOT, "gpurun_out", "bench_1gpu_cache.json")


def cache_key(args, wl, W, H):
    return "%s %dx%d swap%d rot%d %s" % (wl, W, H, args.swap, args.rotation, args.mask)


def cache_1gpu_store(key, mpx, frames):
    """Remember what one GPU did on this workload, for the efficiency figure of a later multi-GPU run here."""
    try:
        os.makedirs(os.path.dirname(CACHE_1GPU), exist_ok=True)
        try:
            with open(CACHE_1GPU) as f:
                c = json.load(f)
        except Exception:
            c = {}
        c[key] = {"Mpx_s": mpx, "frames": frames, "host": socket.gethostname(), "time": time.time()}
        with open(CACHE_1GPU, "w") as f:
            json.dump(c, f)
    except Exception:
        pass  # a read-only tree costs the later run its efficiency figure, nothing else


def cache_1gpu_load(key, max_age_s=6 * 3600):
    try:
        with open(CACHE_1GPU) as f:
            e = json.load(f).get(key)
        if e and e.get("host") == socket.gethostname() and time.time() - e["time"] < max_age_s:
            return float(e["Mpx_s"])
    except Exception:
        pass
    return None


def frames_for(frames_per_gpu, world, one_gpu_default):
    """(frames per GPU, scaling): an explicit per-GPU batch is "weak" scaling; otherwise one GPU runs the workload's
    single-GPU batch and N > 1 GPUs share the 65 536-frame batch (scaled down for workloads with a smaller default)."""
    if frames_per_gpu > 0:
        return frames_per_gpu, "weak"
    if world > 1:
        return max(1, (MULTI_GPU_TOTAL_FRAMES * one_gpu_default // 4096) // world), "strong"
    return one_gpu_default, "weak"


def kernel_sources_sha():
    """Identifies the kernel sources a profile was taken with (profiles/traffic.json carries the same digest)."""
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "vvc_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


# ---------------------------------------------------------------------------------------------------------------------
# N > 1 without a torchrun environment: start the ranks as a child process.  Nothing here may touch the GPU: no
# torch.cuda call that initialises it, no liblvgpu call (the children own the devices).
# ---------------------------------------------------------------------------------------------------------------------
def spawn_command(n, argv, port):
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(n, argv):
    """Runs `bench.py argv` on n ranks of this node and returns the child's exit code."""
    if os.environ.get("LVGPU_BENCH_ONE_DEVICE") != "1":
        import torch  # counting devices does not initialise the GPU
        have = torch.cuda.device_count()
        if have < n:
            print("bench.py: --gpus %d asked for, %d GPU(s) visible" % (n, have), file=sys.stderr)
            return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.run(spawn_command(n, argv, free_port()), env=env).returncode


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
    swap_bytes = 0 if args.swap else 1
    dst = O.synth(O.SYNTH_U16, SEED_DST, 0, n * px)
    src = O.synth(kind, SEED_SRC, 0, n * px)
    out = np.empty(n * px, np.uint16)
    want_out = args.workload != "blend"

    def run(d, s, o, k):
        O.argb_blend_flush_batch(d, s, o, W, H, k, args.rotation, args.swap, threads)
        if o is not None and swap_bytes:  # LV_COLOR_16_SWAP 0 buffers reach the panel byte-swapped (src/display.h:26-30)
            o.byteswap(inplace=True)

    # parity of the first frames (one step applied to fresh inputs)
    chk = dst[: 4 * px].copy()
    chk_out = np.empty(4 * px, np.uint16)
    run(chk, src, chk_out, 4)
    ok = bool(np.array_equal(chk, d0_first_frames_after_one_step))
    if out_first is not None:
        ok = ok and bool(np.array_equal(chk_out, out_first))
    t0 = time.perf_counter()
    run(dst, src, out if want_out else None, n)
    once = time.perf_counter() - t0
    reps = max(1, min(200, int(1.5 / max(once, 1e-4))))
    t0 = time.perf_counter()
    for _ in range(reps):
        run(dst, src, out if want_out else None, n)
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


def eighth_gpu(lvgpu, step_on, n_px, bpp, copy_on=None):
    """The same launch restricted to 32 of the 256 CUs (hipExtStreamCreateWithCUMask), i.e. 1/8 of the GPU's compute.
    The runtime deals CU-mask bits round-robin over the XCDs and rejects a mask that leaves an XCD empty, so the
    32 CUs are 4 per XCD; the memory system behind them is the whole device's: this is NOT a memory partition, and
    the figure to read is frac_of_measured_copy (against what the same 32 CUs reach on a plain copy)."""
    try:
        ms = lvgpu.MaskedStream(range(32))
    except lvgpu.LvgpuError as e:
        return {"error": str(e)}
    census = ms.census()
    import torch
    ts = torch.cuda.ExternalStream(ms.handle.value)  # HIP events on the masked stream itself

    def timed(fn, k=10):
        for _ in range(2):
            fn(ms.handle)
        ms.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(ts)
        for _ in range(k):
            fn(ms.handle)
        e1.record(ts)
        ms.synchronize()
        return e0.elapsed_time(e1) * 1e-3 / k

    dt = timed(step_on)
    copy_gbs = None
    if copy_on is not None:  # the empirical ceiling of the same 32 CUs: a plain device copy (2 B read + 2 B written per px)
        copy_gbs = 4 * n_px / timed(copy_on) / 1e9
    ms.close()
    gbs = bpp * n_px / dt / 1e9
    res = {"cus": 32, "workgroups_per_xcd": census, "value": round(n_px / dt / 1e6, 1), "unit": "Mpx/s",
           "achieved_GBs": round(gbs, 1), "nominal_peak_GBs": HBM_PEAK_GBS / 8,
           "frac_of_nominal": round(gbs / (HBM_PEAK_GBS / 8), 4)}
    if copy_gbs is not None:
        res["measured_copy_GBs"] = round(copy_gbs, 1)
        res["frac_of_measured_copy"] = round(gbs / copy_gbs, 4)
    return res


def profiled_traffic(workload, n, W, H, swap):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/traffic.json), only when they were
    taken for exactly this launch shape with exactly these kernel sources; otherwise (None, why)."""
    tj = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(tj) as f:
            t = json.load(f)
    except Exception:
        return None, "no profiles/traffic.json"
    e = t.get("workloads", {}).get(workload if swap else workload + "_swap0")
    if not isinstance(e, dict):
        return None, "workload not profiled"
    if [n, W, H] != e.get("shape"):
        return None, "profiled for another shape"
    if t.get("kernel_sources_sha") != kernel_sources_sha():
        return None, "kernel sources changed since the profile in %s" % t.get("source", "profiles/")
    return e.get("hbm_bytes_per_launch"), t.get("source", "profiles/")


def main(argv=None):
    args = parse(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:] if argv is None else argv))

    import torch
    from vvc_amd import lvgpu, sharding

    # LVGPU_BENCH_BACKEND=gloo + LVGPU_BENCH_ONE_DEVICE=1 rehearse the multi-rank code path on a single GPU
    # (all ranks share device 0; the numbers are then meaningless, the control flow is what is exercised).
    backend = os.environ.get("LVGPU_BENCH_BACKEND", "gloo")
    one_dev = os.environ.get("LVGPU_BENCH_ONE_DEVICE") == "1"
    if one_dev:
        os.environ["LOCAL_RANK_REAL"] = os.environ.get("LOCAL_RANK", "0")
        os.environ["LOCAL_RANK"] = "0"
    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    if env_world > 1 and not one_dev and torch.cuda.device_count() < env_world:
        sys.exit("bench.py: %d ranks but %d GPU(s) visible" % (env_world, torch.cuda.device_count()))
    # LVGPU_BENCH_FORCE_DIST=1: join a process group even as the only rank (rehearses the RCCL calls on one GPU)
    force_dist = os.environ.get("LVGPU_BENCH_FORCE_DIST") == "1" and "MASTER_ADDR" in os.environ
    rank, world, local_rank, dist = sharding.init(backend if (env_world > 1 or force_dist) else None, force=force_dist)
    if args.gpus > 1 and args.gpus != world:
        sys.exit("bench.py: --gpus %d but the launcher started %d rank(s)" % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    dev = torch.cuda.current_device()
    lvgpu.init(dev)

    wl = args.workload
    bpp, W0, H0, n0 = WORKLOADS[wl]
    W, H = args.width or W0, args.height or H0
    n, scaling = frames_for(args.frames_per_gpu, world, n0)
    px = W * H
    lo, hi = sharding.frame_range(rank, world, n * world)  # this rank's global frame indices
    assert hi - lo == n
    first = lo * px  # global element index of this shard: inputs depend only on the global frame index
    rot = args.rotation
    swap_bytes = 0 if args.swap else 1
    area = (0, 0, W - 1, H - 1)
    mask_kind = lvgpu.SYNTH_MASK_UI if args.mask == "ui" else lvgpu.SYNTH_MASK_UNI
    argb_kind = lvgpu.SYNTH_ARGB_UI if args.mask == "ui" else lvgpu.SYNTH_ARGB_UNI
    px_per_frame = px  # pixels one frame contributes to the metric
    keep = []          # tensors the descriptors point into

    # Where the buffers lie: the device memory has three regions of 96 GB, and a launch that works on buffers in different
    # regions runs faster (vvc_amd/placement.py).  Draw buffers in one, sources and masks in another, panel images in the
    # third, all carved out of one arena whose region boundaries are found by timing first.
    arena, arena_note = None, "buffers wherever the allocator puts them"
    if args.placement == "spread":
        from vvc_amd.placement import RegionArena
        try:
            arena = RegionArena()
            if args.region_map == "auto" or not arena.set_map(args.region_map):
                arena.calibrate(lvgpu)
        except RuntimeError as e:   # too little free memory (another rank or process on this device)
            arena_note += " (no arena: %s)" % str(e).split("\n")[0]

    def new_buf(shape, dtype, region):
        if arena is not None:
            try:
                return arena.empty(shape, dtype, region)
            except RuntimeError:
                pass
        return torch.empty(shape, dtype=dtype, device="cuda")

    def new_synth(kind, seed, first_index, shape, region):
        return lvgpu.synth(kind, seed, first_index, shape, out=new_buf(shape, lvgpu._SYNTH_DTYPE[kind], region))

    out = None
    wants_out = wl in ("blend_flush", "blend_flush_nowb", "blend_flush_rgb565a8", "blend_flush_4k", "blend_flush_466", "flush_180_466", "rotate", "scene")
    if wl == "flush_180_466":
        rot = 2
    if wl.endswith("_argb"):
        dst = new_synth(lvgpu.SYNTH_MASK_UNI, SEED_DST, first * 3, (n, H, W, 3), 0)
        ctx = lvgpu.DrawCtx(dst, color_16_swap=args.swap, screen_transp=1)
    else:
        dst = new_synth(lvgpu.SYNTH_U16, SEED_DST, first, (n, H, W), 0)
        ctx = lvgpu.DrawCtx(dst, color_16_swap=args.swap)
    dsc, blist, rec = None, None, None
    if wl in ("blend_flush", "blend_flush_nowb", "blend", "blend_flush_4k", "blend_flush_466"):
        src = new_synth(argb_kind, SEED_SRC, first, (n, H, W), 1)
        dsc = lvgpu.BlendDsc(area, src_buf=src, opa=255, src_format=lvgpu.SRC_ARGB8888,
                             flags=lvgpu.FLAG_DISCARD_DST if wl == "blend_flush_nowb" else 0)
    elif wl.endswith("rgb565a8"):  # LV_IMG_CF_RGB565A8: colour plane + A8 plane (lv_draw_sw_img.c:85-95)
        src16 = new_synth(lvgpu.SYNTH_U16, SEED_SRC, first, (n, H, W), 1)
        a8 = new_synth(mask_kind, SEED_MASK, first, (n, H, W), 1)
        dsc = lvgpu.BlendDsc(area, src_buf=src16, mask_buf=a8, opa=255)
    elif wl in CLIPPED_AREAS:
        ba = CLIPPED_AREAS[wl]
        bw, bh = ba[2] - ba[0] + 1, ba[3] - ba[1] + 1
        px_per_frame = bw * bh
        src16 = new_synth(lvgpu.SYNTH_U16, SEED_SRC, lo * px_per_frame, (n, bh, bw), 1)
        a8 = new_synth(mask_kind, SEED_MASK, lo * px_per_frame, (n, bh, bw), 1)
        dsc = lvgpu.BlendDsc(ba, src_buf=src16, mask_buf=a8, opa=255)
    elif wl == "fill_translucent":
        # a colour / opacity pair for which lv_color_mix and lv_color_mix_premult disagree on black (about one pair in
        # twenty does), so that the scan-order pre-pass (first non-black pixel of every frame) runs
        color, opa = (0xD67F, 137) if args.swap else (0x2EDE, 103)
        dsc = lvgpu.BlendDsc(area, color=color, opa=opa)
    elif wl == "fill_opaque":
        dsc = lvgpu.BlendDsc(area, color=0x1234, opa=255)
    elif wl == "map_opaque":
        src16 = new_synth(lvgpu.SYNTH_U16, SEED_SRC, first, (n, H, W), 1)
        dsc = lvgpu.BlendDsc(area, src_buf=src16, opa=255)
    elif wl in ("fill_masked", "fill_masked_argb"):
        a8 = new_synth(mask_kind, SEED_MASK, first, (n, H, W), 1)
        dsc = lvgpu.BlendDsc(area, color=0x1234, mask_buf=a8, opa=255)
    elif wl == "scene":
        assert (W, H) == (480, 320), "the recorded scenes are 480x320"
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from util import load_scene
        _fb, rec = load_scene(os.path.join(ROOT, "tests", "golden", "scene2_f0_swap%d.npz" % args.swap))
        dscs = []
        for r in rec:
            s16 = None if r["src"] is None else torch.from_numpy(r["src"].view("int16").copy()).cuda()
            m8 = None if r["mask"] is None else torch.from_numpy(r["mask"].copy()).cuda()
            keep += [s16, m8]
            dscs.append(lvgpu.BlendDsc(r["area"], src_buf=s16, color=r["color"], mask_buf=m8, opa=r["opa"],
                                       blend_mode=r["blend_mode"]))
        blist = lvgpu.BlendList(ctx, dscs).compile(flush=True, rotation=rot, swap_bytes=swap_bytes)  # planned once

    if wants_out:
        out_shape = (n, W, H) if rot in (1, 3) else (n, H, W)
        out = new_buf(out_shape, torch.int16, 2)

    # the descriptor is marshalled once: the clipped-area kernels finish faster than Python fills a ctypes struct
    prepared = None
    if dsc is not None:
        prepared = (lvgpu.PreparedBlend(ctx, dsc, rot, swap_bytes, out) if wants_out else lvgpu.PreparedBlend(ctx, dsc))

    def step(stream=None):
        if blist is not None:
            blist.replay(out=out, stream=stream)
        elif wl in ("rotate", "flush_180_466"):
            lvgpu.flush_rotate(dst, rot, swap_bytes=swap_bytes, out=out, stream=stream)
        else:
            prepared.run(stream)

    # first step on fresh inputs: keep the first frames for the CPU cross-check
    step()
    torch.cuda.synchronize()
    kernel = lvgpu.last_kernel()
    d_first = o_first = None
    if wl in ("blend_flush", "blend", "scene"):
        d_first = dst[:4].cpu().numpy().view("uint16").reshape(-1).copy()
        if wants_out:
            o_first = out[:4].cpu().numpy().view("uint16").reshape(-1).copy()
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
    own_ev_s = ev_ms / 1e3
    dt, ev_ms, ev_median = sharding.max_over_ranks(dist, [dt, ev_ms, ev_median])
    per_rank_dt = sharding.gather_floats(dist, own_ev_s)
    host_dt = dt
    if world > 1:
        dt = ev_ms / 1e3  # the slowest rank's device time for its K steps; the closing barrier's latency stays out

    eighth = None
    copy_gbs = None
    if rank == 0 and world == 1 and not args.no_eighth and wl in ("blend_flush", "blend", "rotate"):
        cp_out = out.view(dst.shape) if out is not None else torch.empty_like(dst)
        # the device's empirical ceiling beside the nominal peak: a plain copy of the draw buffers (2 B read + 2 B
        # written per pixel) through the same library, timed the same way
        cevs = [torch.cuda.Event(enable_timing=True) for _ in range(8)]
        lvgpu.flush_rotate(dst, 0, out=cp_out)
        cevs[0].record()
        for i in range(7):
            lvgpu.flush_rotate(dst, 0, out=cp_out)
            cevs[i + 1].record()
        torch.cuda.synchronize()
        copy_gbs = 4 * px * n / (sorted(cevs[i].elapsed_time(cevs[i + 1]) for i in range(7))[3] / 1e3) / 1e9
        eighth = eighth_gpu(lvgpu, lambda st: step(st.value), px * n, bpp,
                            copy_on=lambda st: lvgpu.flush_rotate(dst, 0, out=cp_out, stream=st.value))
        torch.cuda.synchronize()

    # beside the spread placement: the same launch on buffers the allocator places one after the other (what a host gets
    # that does not care), outside the timed region and only in the full default run
    compact_ms = None
    if rank == 0 and world == 1 and arena is not None and not args.no_eighth and wl in ("blend_flush", "blend", "rotate"):
        try:
            d2 = lvgpu.synth(lvgpu.SYNTH_U16, SEED_DST, first, (n, H, W))
            o2 = torch.empty(out.shape, dtype=torch.int16, device="cuda") if wants_out else None
            if wl == "rotate":
                def fn2():
                    lvgpu.flush_rotate(d2, rot, swap_bytes=swap_bytes, out=o2)
            else:
                s2 = lvgpu.synth(argb_kind, SEED_SRC, first, (n, H, W))
                dsc2 = lvgpu.BlendDsc(area, src_buf=s2, opa=255, src_format=lvgpu.SRC_ARGB8888)
                ctx2 = lvgpu.DrawCtx(d2, color_16_swap=args.swap)
                p2 = lvgpu.PreparedBlend(ctx2, dsc2, rot, swap_bytes, o2) if wants_out else lvgpu.PreparedBlend(ctx2, dsc2)
                fn2 = p2.run
            fn2()
            c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            c0.record()
            for _ in range(10):
                fn2()
            c1.record()
            torch.cuda.synchronize()
            compact_ms = c0.elapsed_time(c1) / 10
        except (RuntimeError, lvgpu.LvgpuError) as e:
            print("bench.py: compact-placement comparison skipped: %s" % e, file=sys.stderr)

    if rank == 0:
        total_px = px_per_frame * n * world * args.steps
        launch_s = ev_ms / 1e3 / args.steps  # HIP events on the launch stream around the timed steps
        achieved = bpp * px_per_frame * n / launch_s / 1e9
        traffic, traffic_source = profiled_traffic(wl, n, W, H, args.swap)
        if blist is not None:
            desc = ("scene: %d LVGL blend descriptors (recorded widget scene) as one compiled display list + flush, "
                    "LV_COLOR_16_SWAP %d, %dx%d logical, rotation %d, %d frames per GPU" % (blist.n, args.swap, W, H, rot * 90, n))
        elif wl in CLIPPED_AREAS:
            ba = CLIPPED_AREAS[wl]
            desc = ("%s: lv_draw_sw_blend of a per-frame RGB565 image + A8 mask (%s) over the clipped area %s of %dx%d "
                    "%sbuffers, opa 255, LV_COLOR_16_SWAP %d, %d frames per GPU; pixels counted = the area's"
                    % (wl, args.mask, list(ba), W, H, "3-byte ARGB8565 (screen_transp) " if wl.endswith("_argb") else "",
                       args.swap, n))
        elif wl.startswith("fill") or wl == "map_opaque":
            desc = ("%s: %s over whole %dx%d buffers, LV_COLOR_16_SWAP %d, %d frames per GPU"
                    % (wl, {"fill_translucent": "fill_normal, no mask, opa %d (premultiplied formula + scan-order pre-pass)" % (137 if args.swap else 103),
                            "fill_opaque": "fill_normal, no mask, opa 255 (lv_color_fill)",
                            "map_opaque": "map_normal, no mask, opa 255 (row copy)",
                            "fill_masked": "fill_normal with an A8 mask (%s), opa 255" % args.mask,
                            "fill_masked_argb": "fill_argb with an A8 mask (%s), opa 255, into a 3-byte ARGB8565 draw "
                                                "buffer (screen_transp)" % args.mask}[wl], W, H, args.swap, n))
        elif wl in ("rotate", "flush_180_466"):
            desc = ("%s: the flush alone (panel image from the draw buffer), LV_COLOR_16_SWAP %d%s, %dx%d logical, rotation %d, "
                    "%d frames per GPU" % (wl, args.swap, " (byte-swapped into wire order)" if swap_bytes else "", W, H, rot * 90, n))
        else:
            desc = ("%s: image with per-pixel alpha over RGB565 (%s alpha), opa 255, LV_COLOR_16_SWAP %d%s, %dx%d logical, "
                    "rotation %d, %d frames per GPU%s"
                    % (wl, args.mask, args.swap, " (flush byte-swaps into wire order)" if swap_bytes and wants_out else "",
                       W, H, rot * 90, n,
                       "; the blended draw buffer is NOT written back (full_refresh display, LVGPU_BLEND_FLAG_DISCARD_DST)"
                       if wl == "blend_flush_nowb" else ""))
        res = {
            "metric": "Mpixels/sec batched 320x480 RGB565 blend+rotate; % of HBM peak at 1/8 GPU",
            "value": round(total_px / dt / 1e6, 1),
            "unit": "Mpx/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "u16",
            "data": "synthetic",
            "config": {
                "workload": desc,
                "frames_per_gpu": n, "total_frames": n * world, "parallelism": "frames sharded, no collective",
                "region_map": arena.map_string() if arena is not None else None,
                "placement": ("draw buffers, sources and panel images in regions 0, 1, 2 of " + arena.describe())
                if arena is not None else arena_note,
            },
            "roofline": {
                "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
                "kernel": kernel, "algorithmic_bytes_per_px": bpp, "launch_ms": round(launch_s * 1e3, 4),
                "launch_ms_median": round(ev_median, 4),
            },
            "pct_hbm_peak": round(100 * achieved / HBM_PEAK_GBS, 2),
        }
        if compact_ms is not None:
            res["roofline"]["compact_placement_launch_ms"] = round(compact_ms, 4)
            res["roofline"]["compact_placement_frac"] = round(bpp * px_per_frame * n / (compact_ms / 1e3) / 1e9 / HBM_PEAK_GBS, 4)
        if copy_gbs is not None:
            res["roofline"]["measured_copy_GBs"] = round(copy_gbs, 1)
            res["roofline"]["frac_of_measured_copy"] = round(achieved / copy_gbs, 4)
        if world > 1:
            res["per_rank_Mpx_s"] = [round(px_per_frame * n * args.steps / t / 1e6, 1) for t in per_rank_dt]
            res["host_ms_per_step"] = round(host_dt / args.steps * 1e3, 4)
            res["timing"] = "max over ranks of the HIP-event time around the K steps (ranks released together by a barrier)"
            res["rank_sync_backend"] = backend
            ref = args.ref_1gpu_mpx or cache_1gpu_load(cache_key(args, wl, W, H))
            if ref:
                res["ref_1gpu_Mpx_s"] = round(ref, 1)
                res["ref_1gpu_source"] = "--ref-1gpu-mpx" if args.ref_1gpu_mpx else "gpurun_out/bench_1gpu_cache.json"
                res["efficiency_vs_1gpu"] = round(res["value"] / (world * ref), 4)
        elif not one_dev:
            cache_1gpu_store(cache_key(args, wl, W, H), res["value"], n)
        if eighth is not None:
            res["eighth_gpu"] = eighth
        if world == 1 and not args.no_cpu_baseline and wl in ("blend_flush", "blend"):
            res["cpu_baseline"] = cpu_baseline(args, W, H, d_first, o_first)
        if world == 1 and not args.no_cpu_baseline and wl == "scene":
            res["cpu_baseline"] = cpu_baseline_scene(args, W, H, rec, d_first, o_first)
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
