"""Loader for liblvgpu.so (built in-tree by vvc_amd/csrc/Makefile).  There is no fallback of any kind: if the
library is missing or a symbol of include/lvgpu.h is absent, importing fails loudly."""
import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "liblvgpu.so")
HEADERS = [os.path.join(_HERE, "..", "include", "lvgpu.h"), os.path.join(_HERE, "..", "include", "lvgpu_lvgl.h")]


class Area(C.Structure):
    """lvgpu_area_t == lv_area_t (libraries/lvgl/src/misc/lv_area.h:43-48)."""
    _fields_ = [("x1", C.c_int16), ("y1", C.c_int16), ("x2", C.c_int16), ("y2", C.c_int16)]

    @classmethod
    def of(cls, a):
        if isinstance(a, cls):
            return a
        x1, y1, x2, y2 = a
        return cls(int(x1), int(y1), int(x2), int(y2))

    @property
    def w(self):
        return self.x2 - self.x1 + 1

    @property
    def h(self):
        return self.y2 - self.y1 + 1

    def tup(self):
        return (self.x1, self.y1, self.x2, self.y2)


class BlendBatch(C.Structure):
    """lvgpu_blend_batch_t"""
    _fields_ = [
        ("dst", C.c_void_p), ("dst_frame_stride", C.c_int64),
        ("buf_area", Area), ("clip_area", Area), ("blend_area", Area),
        ("src", C.c_void_p), ("src_frame_stride", C.c_int64),
        ("mask", C.c_void_p), ("mask_frame_stride", C.c_int64), ("mask_area", Area),
        ("n_frames", C.c_int32), ("src_format", C.c_int32),
        ("color", C.c_uint16), ("mask_res", C.c_uint8), ("opa", C.c_uint8),
        ("blend_mode", C.c_uint8), ("color_16_swap", C.c_uint8), ("dst_argb8565", C.c_uint8), ("flags", C.c_uint8),
    ]


class FlushBatch(C.Structure):
    """lvgpu_flush_batch_t"""
    _fields_ = [
        ("inp", C.c_void_p), ("in_frame_stride", C.c_int64),
        ("out", C.c_void_p), ("out_frame_stride", C.c_int64),
        ("width", C.c_int32), ("height", C.c_int32), ("rotation", C.c_int32), ("swap_bytes", C.c_int32),
        ("n_frames", C.c_int32), ("reserved", C.c_int32),
    ]


class BlendFlushBatch(C.Structure):
    """lvgpu_blend_flush_batch_t"""
    _fields_ = [("blend", BlendBatch), ("out", C.c_void_p), ("out_frame_stride", C.c_int64),
                ("rotation", C.c_int32), ("swap_bytes", C.c_int32)]


def declared_symbols():
    """Every function name declared in include/*.h (used by the tests to check the export table)."""
    names = []
    for h in HEADERS:
        if not os.path.exists(h):
            continue
        text = open(h).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names += re.findall(r"\b(lvgpu_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "liblvgpu.so not found at %s -- build it with `make -C vvc_amd/csrc` (hipcc, gfx950). "
            "There is no CPU fallback." % LIB_PATH)
    # torch brings its own copy of the HIP runtime; when it is to be used in the same process (device memory for the
    # tests and bench.py) it has to be loaded first, or the process ends up with two runtimes of which the second sees
    # no device.  Without torch -- not installed, or installed but unable to load its own libraries -- the library is
    # loaded on its own.
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, u64, sz = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_size_t
    sig = {
        "lvgpu_init": (C.c_int, [C.c_int]),
        "lvgpu_version": (C.c_int, []),
        "lvgpu_last_error": (C.c_char_p, []),
        "lvgpu_last_kernel": (C.c_char_p, []),
        "lvgpu_device_count": (C.c_int, []),
        "lvgpu_malloc": (C.c_int, [C.POINTER(vp), sz]),
        "lvgpu_free": (C.c_int, [vp]),
        "lvgpu_memcpy_h2d": (C.c_int, [vp, vp, sz, vp]),
        "lvgpu_memcpy_d2h": (C.c_int, [vp, vp, sz, vp]),
        "lvgpu_memset": (C.c_int, [vp, C.c_int, sz, vp]),
        "lvgpu_stream_synchronize": (C.c_int, [vp]),
        "lvgpu_blend_batch": (C.c_int, [C.POINTER(BlendBatch), vp]),
        "lvgpu_blend_batch_noalloc": (C.c_int, [C.POINTER(BlendBatch), vp]),
        "lvgpu_flush_rotate_batch": (C.c_int, [C.POINTER(FlushBatch), vp]),
        "lvgpu_blend_flush_batch": (C.c_int, [C.POINTER(BlendFlushBatch), vp]),
        "lvgpu_flush_chunks": (C.c_int, [C.POINTER(Area), C.c_int32, C.c_int32, C.c_int32, C.c_uint32,
                                         C.POINTER(FlushChunk), C.c_int32]),
        "lvgpu_axs15231b_transactions": (C.c_int, [C.POINTER(FlushChunk), C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                                   C.POINTER(PanelTxn)]),
        "lvgpu_blend_list_batch": (C.c_int, [C.POINTER(BlendBatch), C.c_int32, vp]),
        "lvgpu_list_create": (C.c_int, [C.POINTER(BlendBatch), C.POINTER(BlendBatch), C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                        C.POINTER(vp)]),
        "lvgpu_list_replay": (C.c_int, [vp, vp, C.c_int32, vp, C.c_int64, vp]),
        "lvgpu_list_destroy": (C.c_int, [vp]),
        "lvgpu_blend_list_plan": (C.c_int, [C.POINTER(BlendBatch), C.POINTER(BlendBatch), C.c_int32, C.c_int32, C.c_int32,
                                            C.POINTER(ListPlan)]),
        "lvgpu_blend_list_flush_batch": (C.c_int, [C.POINTER(BlendBatch), C.POINTER(BlendBatch), C.c_int32, vp, C.c_int64,
                                                   C.c_int32, C.c_int32, vp]),
        "lvgpu_color_from_rgb888": (C.c_int, [vp, vp, sz, C.c_int, vp]),
        "lvgpu_color_to32": (C.c_int, [vp, vp, sz, C.c_int, vp]),
        "lvgpu_image_split": (C.c_int, [vp, C.c_int, vp, vp, sz, C.c_int, vp]),
        "lvgpu_image_convert": (C.c_int, [vp, C.c_int, vp, vp, sz, C.c_int, C.POINTER(ImageOpts), vp]),
        "lvgpu_color_fill": (C.c_int, [vp, C.c_uint16, sz, vp]),
        "lvgpu_buffer_copy_batch": (C.c_int, [vp, i64, i32, Area, vp, i64, i32, Area, i32, vp]),
        "lvgpu_synth": (C.c_int, [vp, C.c_int, u64, u64, sz, vp]),
        "lvgpu_checksum": (C.c_int, [vp, C.c_int, u64, sz, vp, vp]),
        "lvgpu_stream_create_cu_mask": (C.c_int, [C.POINTER(vp), vp, C.c_int]),
        "lvgpu_stream_create": (C.c_int, [C.POINTER(vp)]),
        "lvgpu_stream_destroy": (C.c_int, [vp]),
        "lvgpu_xcc_census": (C.c_int, [vp, C.c_int, vp]),
        "lvgpu_debug_counters": (None, [C.POINTER(DebugCounters)]),
        "lvgpu_lvgl_bind": (None, [vp, vp, vp]),
        "lvgpu_lvgl_set_color_16_swap": (None, [C.c_int]),
        "lvgpu_lvgl_set_deferred": (None, [C.c_int]),
        "lvgpu_lvgl_inject_failure": (None, [C.c_int]),
        "lvgpu_draw_ctx_init": (None, [vp, vp]),
        "lvgpu_draw_ctx_deinit": (None, [vp, vp]),
        "lvgpu_draw_ctx_size": (sz, []),
        "lvgpu_blend": (None, [vp, vp]),
        "lvgpu_wait_for_finish": (None, [vp]),
        "lvgpu_flush_cb": (None, [vp, vp, vp]),
        "lvgpu_lvgl_buf_alloc": (vp, [sz]),
        "lvgpu_lvgl_buf_free": (None, [vp]),
        "lvgpu_lvgl_get_stats": (None, [vp]),
        "lvgpu_lvgl_reset_stats": (None, []),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)  # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


class ListPlan(C.Structure):
    """lvgpu_list_plan_t"""
    _fields_ = [(n, C.c_int32) for n in ("tiles", "tiles_selected", "tiles_overwritten", "items_listed", "items_standalone",
                                         "items_noop", "launches", "flush_fused")] + \
               [(n, C.c_int64) for n in ("item_steps", "batch_steps", "chunks_full", "chunks_partial", "px_item_steps",
                                         "px_chunks", "px_culled")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class ImageOpts(C.Structure):
    """lvgpu_image_opts_t"""
    _fields_ = [("chroma_keyed", C.c_int32), ("chroma_key", C.c_uint16), ("recolor", C.c_uint16), ("recolor_opa", C.c_uint8),
                ("reserved", C.c_uint8 * 3)]


class DebugCounters(C.Structure):
    """lvgpu_debug_counters_t"""
    _fields_ = [("alloc_calls", C.c_uint64), ("first_nonblack_launches", C.c_uint64), ("band_pad_bytes", C.c_uint64)]


class FlushChunk(C.Structure):
    """lvgpu_flush_chunk_t"""
    _fields_ = [("x_start", C.c_int32), ("y_start", C.c_int32), ("x_end", C.c_int32), ("y_end", C.c_int32),
                ("offset_px", C.c_int64), ("n_px", C.c_int32), ("reserved", C.c_int32)]


class PanelTxn(C.Structure):
    """lvgpu_panel_txn_t"""
    _fields_ = [("cmd", C.c_uint32), ("param_len", C.c_uint32), ("param", C.c_uint8 * 4), ("is_color", C.c_int32),
                ("data_offset_px", C.c_int64), ("data_len", C.c_int64)]


class LvgpuError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("lvgpu error %d: %s" % (code, msg))
        self.code = code


def check(rc):
    if rc != 0:
        raise LvgpuError(rc, lib().lvgpu_last_error().decode())
    return rc
