"""Live LVGL (oracle/_ref, the real library) rendering one widget scene per refresh: LVGL's software blend, the GPU
plug-in in immediate mode (one staged blend per call) and in deferred mode (one display list per flush).
Run on the GPU box:  python tests/perf/bench_dropin.py"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import lvref_py as O  # noqa: E402
from vvc_amd import _lib, lvgpu  # noqa: E402


def addr(fn):
    return C.cast(fn, C.c_void_p)


def main():
    W, H, scene, reps = 480, 320, 2, 20
    flags = int(sys.argv[1]) if len(sys.argv) > 1 else 0  # ref_scene_display flags: 1 = AA off, 2 = transparent screen
    bpp = 3 if flags & 2 else 2
    lvgpu.init(0)
    R = O.ref(1)
    L = _lib.lib()
    L.lvgpu_lvgl_bind(addr(R.lv_draw_sw_init_ctx), addr(R.lv_draw_sw_blend_basic), addr(R.lv_disp_flush_ready))
    L.lvgpu_lvgl_set_color_16_swap(1)
    frames = {}

    class FlushCfg(C.Structure):  # lvgpu_flush_cfg_t
        _fields_ = [("sw_rotate", C.c_int32), ("trans_size", C.c_uint32), ("swap_bytes", C.c_int32),
                    ("draw_bitmap", C.c_void_p), ("draw_wait", C.c_void_p), ("user", C.c_void_p)]

    sink = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint16))(lambda *a: None)
    cfg = FlushCfg(1, W * H // 10, 0, C.cast(sink, C.c_void_p), None, None)
    modes = ["lvgl_sw", "gpu_immediate", "gpu_deferred_staged", "gpu_deferred"] + \
            ([] if flags & 2 else ["gpu_deferred_staged+flush_cb", "gpu_deferred+flush_cb"])
    for mode in modes:
        # deferred mode draws a draw buffer from lvgpu_lvgl_buf_alloc (pinned memory) in place in host memory; one from
        # calloc ("staged") travels to the device and back
        if mode.startswith("gpu_deferred") and "staged" not in mode:
            R.ref_scene_set_buf_allocator(addr(L.lvgpu_lvgl_buf_alloc), addr(L.lvgpu_lvgl_buf_free))
        else:
            R.ref_scene_set_buf_allocator(None, None)
        L.lvgpu_lvgl_set_deferred(1 if mode.startswith("gpu_deferred") else 0)
        R.ref_scene_set_pre_flush(addr(L.lvgpu_wait_for_finish) if mode in ("gpu_deferred", "gpu_deferred_staged") else None)
        if mode == "lvgl_sw":
            R.ref_scene_display(W, H, None, None, 0, None, None, 1, flags)
        elif mode.endswith("+flush_cb"):  # blend and 90-degree flush both through the plug-in, ten chunks to a sink
            R.ref_scene_display(W, H, addr(L.lvgpu_draw_ctx_init), addr(L.lvgpu_draw_ctx_deinit), L.lvgpu_draw_ctx_size(),
                                addr(L.lvgpu_flush_cb), C.addressof(cfg), 1, flags)
        else:
            R.ref_scene_display(W, H, addr(L.lvgpu_draw_ctx_init), addr(L.lvgpu_draw_ctx_deinit), L.lvgpu_draw_ctx_size(),
                                None, None, 1, flags)
        R.ref_scene_build(scene)
        R.ref_scene_refresh()
        R.ref_scene_build(scene)
        R.ref_scene_refresh()            # deferred mode: the second identical refresh compiles the list
        L.lvgpu_lvgl_reset_stats()
        t0 = time.perf_counter()
        for _ in range(reps):
            R.ref_scene_build(scene)   # invalidates the screen again
            R.ref_scene_refresh()
        dt = (time.perf_counter() - t0) / reps
        fb = R.ref_scene_draw_buf() if mode.endswith("flush_cb") else R.ref_scene_fb()  # no capture with a foreign flush_cb
        frames[mode] = np.ctypeslib.as_array(C.cast(fb, C.POINTER(C.c_uint8)), shape=(H * W * bpp,)).copy()
        st = (C.c_uint64 * 8)()  # lvgpu_lvgl_stats_t
        L.lvgpu_lvgl_get_stats(st)
        print("%-29s %8.3f ms per refresh of scene %d (%dx%d, flags %d); of it submitting %.3f ms (%d lists, %d replayed, "
              "%d in place)" % (mode, dt * 1e3, scene, W, H, flags, st[6] / reps * 1e-6, st[4], st[5], st[7]), flush=True)
    R.ref_scene_set_buf_allocator(None, None)
    R.ref_scene_display(16, 16, None, None, 0, None, None, 1, 0)
    R.ref_scene_set_pre_flush(None)
    print("identical frames:", all(np.array_equal(frames["lvgl_sw"], v) for v in frames.values()))


if __name__ == "__main__":
    main()
