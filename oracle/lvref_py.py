"""ctypes bindings for the CPU oracle (oracle/liblvref.so) and, where it was built, for the real LVGL compiled from
the reference sources (oracle/_ref/liblvgl_swap{0,1}.so).  TEST INFRASTRUCTURE ONLY.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

MASK_RES_TRANSP, MASK_RES_FULL_COVER, MASK_RES_CHANGED, MASK_RES_UNKNOWN = 0, 1, 2, 3
SRC_FILL, SRC_RGB565, SRC_ARGB8888, SRC_RGBA8888, SRC_ARGB8565 = 0, 1, 2, 3, 4
SYNTH_U16, SYNTH_MASK_UI, SYNTH_MASK_UNI, SYNTH_ARGB_UI, SYNTH_ARGB_UNI = 0, 1, 2, 3, 4


class Area(C.Structure):
    _fields_ = [("x1", C.c_int16), ("y1", C.c_int16), ("x2", C.c_int16), ("y2", C.c_int16)]

    @classmethod
    def of(cls, a):
        if isinstance(a, cls):
            return a
        return cls(*[int(v) for v in a])

    def tup(self):
        return (self.x1, self.y1, self.x2, self.y2)

    @property
    def w(self):
        return self.x2 - self.x1 + 1

    @property
    def h(self):
        return self.y2 - self.y1 + 1


class BlendDsc(C.Structure):
    _fields_ = [
        ("buf", C.c_void_p),
        ("buf_area", Area),
        ("clip_area", Area),
        ("blend_area", Area),
        ("src_buf", C.c_void_p),
        ("color", C.c_uint16),
        ("mask_buf", C.c_void_p),
        ("mask_res", C.c_uint8),
        ("mask_area", Area),
        ("opa", C.c_uint8),
        ("blend_mode", C.c_uint8),
        ("swap", C.c_uint8),
        ("screen_transp", C.c_uint8),
        ("antialiasing_off", C.c_uint8),
    ]


def build(force=False):
    """Compile oracle/liblvref.so (always possible) and oracle/_ref (only where /root/reference exists)."""
    so = os.path.join(_HERE, "liblvref.so")
    src_newer = (not os.path.exists(so)) or any(
        os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(so) for f in ("lvref.c", "lvref.h")
    )
    if force or src_newer:
        subprocess.check_call(["make", "-C", _HERE, "liblvref.so"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference/libraries/lvgl/src"):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is None:
        so = os.path.join(_HERE, "liblvref.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        L.lvref_color_mix.restype = C.c_uint16
        L.lvref_color_mix.argtypes = [C.c_uint16, C.c_uint16, C.c_uint8, C.c_int]
        L.lvref_color_mix_premult.restype = C.c_uint16
        L.lvref_color_mix_premult.argtypes = [C.c_void_p, C.c_uint16, C.c_uint8, C.c_int]
        L.lvref_color_premult.argtypes = [C.c_uint16, C.c_uint8, C.c_void_p, C.c_int]
        L.lvref_color_make.restype = C.c_uint16
        L.lvref_color_make.argtypes = [C.c_uint8, C.c_uint8, C.c_uint8, C.c_int]
        L.lvref_color_hex.restype = C.c_uint16
        L.lvref_color_hex.argtypes = [C.c_uint32, C.c_int]
        L.lvref_color_to32.restype = C.c_uint32
        L.lvref_color_to32.argtypes = [C.c_uint16, C.c_int]
        L.lvref_color_mix_n.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        L.lvref_color_mix_digest.restype = C.c_uint64
        L.lvref_color_mix_digest.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.c_int]
        L.lvref_color_hex_n.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        L.lvref_color_to32_n.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        L.lvref_color_mix_with_alpha.argtypes = [C.c_uint16, C.c_uint8, C.c_uint16, C.c_uint8, C.c_void_p,
                                                 C.c_void_p, C.c_int]
        L.lvref_blend.restype = C.c_int
        L.lvref_blend.argtypes = [C.POINTER(BlendDsc)]
        L.lvref_image_split.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        L.lvref_image_convert.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int,
                                          C.c_uint16, C.c_uint16, C.c_uint8]
        L.lvref_list_flush_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                             C.c_int]
        L.lvref_img_blend.restype = C.c_int
        L.lvref_img_blend.argtypes = [C.c_void_p, Area, Area, Area, C.c_void_p, C.c_int, C.c_uint8, C.c_int, C.c_int]
        L.lvref_img_blend_aa.restype = C.c_int
        L.lvref_img_blend_aa.argtypes = [C.c_void_p, Area, Area, Area, C.c_void_p, C.c_int, C.c_uint8, C.c_int, C.c_int, C.c_int]
        L.lvref_flush.restype = C.c_int
        L.lvref_flush.argtypes = [C.POINTER(Area), C.c_void_p, C.c_int, C.c_uint32, C.c_int, C.c_int, C.c_void_p,
                                  C.c_void_p]
        L.lvref_axs15231b_draw_bitmap.restype = None
        L.lvref_axs15231b_draw_bitmap.argtypes = [C.c_int] * 8 + [C.c_void_p, C.c_void_p]
        L.lvref_flush_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.lvref_synth.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_uint64, C.c_size_t]
        L.lvref_checksum.restype = C.c_uint64
        L.lvref_checksum.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_size_t]
        L.lvref_argb_blend_flush_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                                   C.c_int, C.c_int, C.c_int]
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def blend(buf, buf_area, clip_area, blend_area, src=None, color=0, mask=None, mask_res=MASK_RES_CHANGED,
          mask_area=None, opa=255, blend_mode=0, swap=1, screen_transp=0, antialiasing=1):
    """In-place lv_draw_sw_blend on one numpy frame (uint16 HxW, or uint8 HxWx3 for screen_transp).  With
    antialiasing=0 the mask array is rounded in place as well, as the reference does (lv_draw_sw_blend.c:150-156)."""
    assert buf.flags.c_contiguous
    d = BlendDsc()
    d.buf = _p(buf)
    d.buf_area = Area.of(buf_area)
    d.clip_area = Area.of(clip_area)
    d.blend_area = Area.of(blend_area)
    d.src_buf = _p(src)
    d.color = int(color)
    d.mask_buf = _p(mask)
    d.mask_res = mask_res
    d.mask_area = Area.of(mask_area if mask_area is not None else blend_area)
    d.opa = int(opa)
    d.blend_mode = blend_mode
    d.swap = swap
    d.screen_transp = screen_transp
    d.antialiasing_off = 0 if antialiasing else 1
    return lib().lvref_blend(C.byref(d))


def image_split(src, fmt, swap):
    n = src.size // (3 if fmt == SRC_ARGB8565 else 4) if src.dtype == np.uint8 else src.size
    color = np.empty(n, np.uint16)
    alpha = np.empty(n, np.uint8)
    lib().lvref_image_split(_p(np.ascontiguousarray(src)), fmt, _p(color), _p(alpha), n, swap)
    return color, alpha


def image_convert(src, fmt, swap, chroma_keyed=0, chroma_key=0, recolor=0, recolor_opa=0):
    """convert_cb + recolor of lv_draw_sw_img_decoded: (colour plane, alpha plane) of an image."""
    src = np.ascontiguousarray(src)
    n = src.size // (3 if fmt == SRC_ARGB8565 else 1)
    color = np.empty(n, np.uint16)
    alpha = np.empty(n, np.uint8)
    lib().lvref_image_convert(_p(src), fmt, _p(color), _p(alpha), n, swap, chroma_keyed, chroma_key, recolor, recolor_opa)
    return color, alpha


def img_blend(buf, buf_area, clip_area, coords, src, fmt, opa, swap, hor_res, antialiasing=1):
    return lib().lvref_img_blend_aa(_p(buf), Area.of(buf_area), Area.of(clip_area), Area.of(coords), _p(src), fmt,
                                    int(opa), swap, hor_res, 0 if antialiasing else 1)


_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint16))


def flush(area, color_map, rotation, trans_size, hor_res, ver_res):
    """Run the restated flush callback; returns the list of (x0, y0, x1, y1, pixels) chunks it sent."""
    chunks = []

    def cb(_user, x0, y0, x1, y1, data):
        n = (x1 - x0) * (y1 - y0)
        chunks.append((x0, y0, x1, y1, np.ctypeslib.as_array(data, shape=(n,)).copy()))

    a = Area.of(area)
    cbo = _CB(cb)
    lib().lvref_flush(C.byref(a), _p(color_map), rotation, trans_size, hor_res, ver_res, C.cast(cbo, C.c_void_p), None)
    return chunks


_TX = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_uint8), C.c_size_t)


def axs15231b_draw_bitmap(x_start, y_start, x_end, y_end, x_gap=0, y_gap=0, qspi=True, bits_per_pixel=16):
    """The restated panel driver's draw_bitmap: [(is_color, lcd_cmd, param bytes, size)] as handed to the IO layer."""
    txs = []

    def tx(_user, is_color, cmd, param, size):
        txs.append((is_color, cmd & 0xFFFFFFFF, b"" if is_color else bytes(param[:size]), size))

    cbo = _TX(tx)
    lib().lvref_axs15231b_draw_bitmap(x_start, y_start, x_end, y_end, x_gap, y_gap, 1 if qspi else 0, bits_per_pixel,
                                      C.cast(cbo, C.c_void_p), None)
    return txs


def flush_frame(inp, W, H, rotation, swap_bytes=0):
    out = np.empty(W * H, np.uint16)
    lib().lvref_flush_frame(_p(np.ascontiguousarray(inp)), _p(out), W, H, rotation, swap_bytes)
    return out


_SYNTH_DT = {SYNTH_U16: np.uint16, SYNTH_MASK_UI: np.uint8, SYNTH_MASK_UNI: np.uint8, SYNTH_ARGB_UI: np.uint32,
             SYNTH_ARGB_UNI: np.uint32}


def synth(kind, seed, first_index, count):
    out = np.empty(count, _SYNTH_DT[kind])
    lib().lvref_synth(_p(out), kind, seed, first_index, count)
    return out


def checksum(arr, first_index=0):
    arr = np.ascontiguousarray(arr)
    return int(lib().lvref_checksum(_p(arr), arr.dtype.itemsize, first_index, arr.size))


def argb_blend_flush_batch(dst, src, out, width, height, n_frames, rotation, swap, threads):
    lib().lvref_argb_blend_flush_batch(_p(dst), _p(src), _p(out), width, height, n_frames, rotation, swap, threads)


# ---------------------------------------------------------------------------------------------------------------
# The real LVGL (oracle/_ref), present only where it was built from /root/reference.
# ---------------------------------------------------------------------------------------------------------------
_ref = {}


class RecItem(C.Structure):
    """ref_rec_t of oracle/ref_harness.c"""
    _fields_ = [("area", C.c_int16 * 4), ("src", C.POINTER(C.c_uint16)), ("mask", C.POINTER(C.c_uint8)),
                ("color", C.c_uint16), ("opa", C.c_uint8), ("blend_mode", C.c_uint8)]


def prepare_list(items, width, height, swap):
    """Marshal a display list once: items are dicts area/src/mask/color/opa/blend_mode (as tests/util.load_scene
    returns them).  Returns an opaque object for list_flush_batch."""
    arr = (BlendDsc * len(items))()
    keep = []
    for i, it in enumerate(items):
        d = arr[i]
        d.buf_area = Area(0, 0, width - 1, height - 1)
        d.clip_area = d.blend_area = d.mask_area = Area.of(it["area"])
        s = None if it["src"] is None else np.ascontiguousarray(it["src"])
        m = None if it["mask"] is None else np.ascontiguousarray(it["mask"])
        keep += [s, m]
        d.src_buf, d.mask_buf = _p(s), _p(m)
        d.color, d.opa, d.blend_mode = int(it["color"]), int(it["opa"]), int(it["blend_mode"])
        d.mask_res, d.swap, d.screen_transp = MASK_RES_CHANGED, swap, 0
    return arr, keep, width, height


def list_flush_batch(dst, prepared, out, n_frames, rotation, threads):
    """The list applied in order to every frame of dst ([n, H, W] uint16, in place), each frame then flushed with the
    rotation into out (flat uint16) unless out is None; OpenMP over frames."""
    arr, _keep, width, height = prepared
    lib().lvref_list_flush_batch(_p(dst), arr, len(arr), _p(out), width, height, n_frames, rotation, threads)


def ref_available(swap=1):
    return os.path.exists(os.path.join(_HERE, "_ref", "liblvgl_swap%d.so" % swap))


def ref(swap=1):
    """The real LVGL build with LV_COLOR_16_SWAP == swap, or None if oracle/_ref was not built."""
    if swap not in _ref:
        # LVREF_REF_DIR: another build of the same recipe (e.g. other optimisation flags for a baseline measurement)
        so = os.path.join(os.environ.get("LVREF_REF_DIR", os.path.join(_HERE, "_ref")), "liblvgl_swap%d.so" % swap)
        if not os.path.exists(so):
            _ref[swap] = None
            return None
        # RTLD_LOCAL: the two builds define the same symbols and must not see each other.
        L = C.CDLL(so, mode=os.RTLD_LOCAL | os.RTLD_NOW)
        L.ref_init.argtypes = [C.c_int, C.c_int]
        L.ref_blend.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint16, C.c_void_p,
                                C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.ref_img_decoded.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                      C.c_uint16, C.c_int, C.c_int]
        L.ref_color_mix_n.argtypes = [C.c_void_p] * 4 + [C.c_size_t]
        L.ref_color_mix_digest.restype = C.c_uint64
        L.ref_color_mix_digest.argtypes = [C.c_int, C.c_uint32, C.c_uint32]
        L.ref_color_premult_n.argtypes = [C.c_void_p] * 3 + [C.c_size_t]
        L.ref_color_mix_premult_n.argtypes = [C.c_void_p] * 4 + [C.c_size_t]
        L.ref_color_hex_n.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.ref_color_make_n.argtypes = [C.c_void_p] * 4 + [C.c_size_t]
        L.ref_color_to32_n.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.ref_color_fill.argtypes = [C.c_void_p, C.c_uint16, C.c_uint32]
        L.ref_color_mix_with_alpha_n.argtypes = [C.c_void_p] * 6 + [C.c_size_t]
        L.ref_buffer_copy.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.ref_layout.restype = C.c_long
        L.ref_layout.argtypes = [C.c_char_p]
        L.ref_disp_drv_bits.restype = C.c_uint32
        L.ref_disp_drv_bits.argtypes = [C.c_char_p]
        L.ref_scene_display.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p,
                                        C.c_int, C.c_int]
        L.ref_scene_fb.restype = C.c_void_p
        L.ref_scene_draw_buf.restype = C.c_void_p
        if hasattr(L, "ref_scene_draw_buf_act"):
            L.ref_scene_draw_buf_act.restype = C.c_void_p
        L.ref_scene_drv.restype = C.c_void_p
        L.ref_scene_build.argtypes = [C.c_int]
        if hasattr(L, "ref_set_antialiasing"):
            L.ref_set_antialiasing.argtypes = [C.c_int]
        if hasattr(L, "ref_scene_set_pre_flush"):
            L.ref_scene_set_pre_flush.argtypes = [C.c_void_p]
        if hasattr(L, "ref_scene_set_buf_allocator"):
            L.ref_scene_set_buf_allocator.argtypes = [C.c_void_p, C.c_void_p]
        if hasattr(L, "ref_record_items"):
            L.ref_record_items.restype = C.POINTER(RecItem)
            L.ref_record_ctx_size.restype = C.c_size_t
        assert L.ref_color_16_swap() == swap
        L.ref_init(480, 480)
        _ref[swap] = L
    return _ref[swap]


def render_scene(L, hor_res, ver_res, scene, flags=0):
    """Widget scene `scene` rendered by the real LVGL on a hor_res x ver_res (physical) display; flags as
    ref_scene_display (bit 2 + bits 3-4: LVGL's own software rotation).  Returns the frame buffer [ver_res, hor_res]."""
    assert L.ref_scene_display(hor_res, ver_res, None, None, 0, None, None, 0 if flags & 4 else 1, flags) == 0
    L.ref_scene_build(scene)
    L.ref_scene_refresh()
    return np.ctypeslib.as_array(C.cast(L.ref_scene_fb(), C.POINTER(C.c_uint16)), shape=(ver_res, hor_res)).copy()


def recorded_scene(L, w, h, scene, flags=0):
    """Render widget scene `scene` with the real LVGL while recording every blend descriptor its producers issue.
    Returns (frame buffer as LVGL drew it [h, w] uint16, list of dicts area/src/mask/color/opa/blend_mode)."""
    vp = lambda f: C.cast(f, C.c_void_p)
    L.ref_record_reset()
    assert L.ref_scene_display(w, h, vp(L.ref_record_ctx_init), vp(L.ref_record_ctx_deinit), L.ref_record_ctx_size(),
                               None, None, 1, flags) == 0
    L.ref_scene_build(scene)
    L.ref_record_reset()
    L.ref_scene_refresh()
    fb = np.ctypeslib.as_array(C.cast(L.ref_scene_fb(), C.POINTER(C.c_uint16)), shape=(h, w)).copy()
    items = []
    rec = L.ref_record_items()
    for i in range(L.ref_record_count()):
        r = rec[i]
        x1, y1, x2, y2 = r.area
        n = (x2 - x1 + 1, ) and ((y2 - y1 + 1), (x2 - x1 + 1))
        src = np.ctypeslib.as_array(r.src, shape=n).copy() if r.src else None
        mask = np.ctypeslib.as_array(r.mask, shape=n).copy() if r.mask else None
        items.append(dict(area=(x1, y1, x2, y2), src=src, mask=mask, color=r.color, opa=r.opa, blend_mode=r.blend_mode))
    L.ref_record_reset()
    L.ref_scene_display(16, 16, None, None, 0, None, None, 1, 0)
    return fb, items


def _area16(a):
    return np.array(Area.of(a).tup(), np.int16)


def ref_blend(L, buf, buf_area, clip_area, blend_area, src=None, color=0, mask=None, mask_res=MASK_RES_CHANGED,
              mask_area=None, opa=255, blend_mode=0, screen_transp=0):
    """lv_draw_sw_blend of the real LVGL on one numpy frame, in place.  The mask is copied first because LVGL may
    threshold it in place."""
    ba, ca, bla = _area16(buf_area), _area16(clip_area), _area16(blend_area)
    ma = _area16(mask_area if mask_area is not None else blend_area)
    m = None if mask is None else mask.copy()
    L.ref_blend(_p(buf), _p(ba), _p(ca), _p(bla), _p(src), int(color), _p(m), mask_res,
                _p(ma) if m is not None else None, int(opa), blend_mode, screen_transp)
