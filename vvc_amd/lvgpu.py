"""Host-side mirror of the reference's drawing interface for the batched GPU path.

The names follow LVGL (libraries/lvgl/src/draw/lv_draw.h:59-73, libraries/lvgl/src/draw/sw/lv_draw_sw_blend.h:29-39):
a DrawCtx is lv_draw_ctx_t {buf, buf_area, clip_area} whose buf is a BATCH of frames in device memory, a BlendDsc is
lv_draw_sw_blend_dsc_t, lv_draw_sw_blend(ctx, dsc) has the argument meaning of the reference function, and
flush_rotate() is the pixel movement of lvgl_port_flush_callback (src/lv_port.c:428-572).  Everything goes through
the C ABI of liblvgpu.so; torch is used only for device memory and streams.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import Area, BlendBatch, BlendFlushBatch, FlushBatch, LvgpuError, check  # noqa: F401

LV_OPA_TRANSP, LV_OPA_MIN, LV_OPA_MAX, LV_OPA_COVER = 0, 2, 253, 255
LV_DRAW_MASK_RES_TRANSP, LV_DRAW_MASK_RES_FULL_COVER, LV_DRAW_MASK_RES_CHANGED, LV_DRAW_MASK_RES_UNKNOWN = 0, 1, 2, 3
LV_BLEND_MODE_NORMAL, LV_BLEND_MODE_ADDITIVE, LV_BLEND_MODE_SUBTRACTIVE, LV_BLEND_MODE_MULTIPLY, \
    LV_BLEND_MODE_REPLACE = 0, 1, 2, 3, 4
LV_DISP_ROT_NONE, LV_DISP_ROT_90, LV_DISP_ROT_180, LV_DISP_ROT_270 = 0, 1, 2, 3
SRC_FILL, SRC_RGB565, SRC_ARGB8888, SRC_RGBA8888, SRC_ARGB8565 = 0, 1, 2, 3, 4
SYNTH_U16, SYNTH_MASK_UI, SYNTH_MASK_UNI, SYNTH_ARGB_UI, SYNTH_ARGB_UNI = 0, 1, 2, 3, 4
FLAG_NO_AA, FLAG_DISCARD_DST = 0x01, 0x04   # LVGPU_BLEND_FLAG_*

_SRC_ELEM_BYTES = {SRC_RGB565: 2, SRC_ARGB8888: 4, SRC_RGBA8888: 4, SRC_ARGB8565: 3}


def init(device=0):
    check(_lib.lib().lvgpu_init(device))


def last_kernel():
    return _lib.lib().lvgpu_last_kernel().decode()


def _stream(stream):
    if stream is None:
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)
    if isinstance(stream, torch.cuda.Stream):
        return C.c_void_p(stream.cuda_stream)
    if isinstance(stream, C.c_void_p):
        return stream
    return C.c_void_p(int(stream))


def _dev(t, what):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise TypeError("%s must be a CUDA/HIP tensor" % what)
    return t


def _frame_stride_bytes(t, n_frames, frame_dims):
    """Byte distance between frames of a tensor that holds either one shared image or n_frames images."""
    if t.dim() == frame_dims or t.shape[0] == 1:
        return 0
    if t.shape[0] != n_frames:
        raise ValueError("tensor holds %d frames, the context %d" % (t.shape[0], n_frames))
    return t.stride(0) * t.element_size()


class DrawCtx:
    """lv_draw_ctx_t over a batch: buf is [n_frames, H, W] 16-bit pixels in device memory."""

    def __init__(self, buf, buf_area=None, clip_area=None, color_16_swap=1, screen_transp=0):
        _dev(buf, "buf")
        self.screen_transp = 1 if screen_transp else 0
        if self.screen_transp:  # ARGB8565 destination: [n_frames, H, W, 3] bytes (lv_draw_sw_blend.c:129-133)
            if buf.dim() != 4 or buf.shape[3] != 3 or buf.element_size() != 1:
                raise ValueError("with screen_transp buf must be [n_frames, H, W, 3] bytes")
        elif buf.dim() != 3 or buf.element_size() != 2:
            raise ValueError("buf must be [n_frames, H, W] with 2-byte elements")
        if buf.shape[0] and not buf[0].is_contiguous():
            raise ValueError("each frame must be contiguous")
        self.buf = buf
        n, h, w = buf.shape[:3]
        self.buf_area = Area.of(buf_area) if buf_area is not None else Area(0, 0, w - 1, h - 1)
        if self.buf_area.w != w or self.buf_area.h != h:
            raise ValueError("buf_area does not match the buffer shape")
        self.clip_area = Area.of(clip_area) if clip_area is not None else self.buf_area
        self.color_16_swap = 1 if color_16_swap else 0

    @property
    def n_frames(self):
        return self.buf.shape[0]


class BlendDsc:
    """lv_draw_sw_blend_dsc_t.  src_buf: [h, w] or [n_frames, h, w] (2- or 4-byte elements), or [.., h, w, 3] uint8
    for SRC_ARGB8565; mask_buf: [h, w] or [n_frames, h, w] uint8."""

    def __init__(self, blend_area, src_buf=None, color=0, mask_buf=None, mask_res=LV_DRAW_MASK_RES_CHANGED,
                 mask_area=None, opa=LV_OPA_COVER, blend_mode=LV_BLEND_MODE_NORMAL, src_format=None, flags=0):
        self.blend_area = Area.of(blend_area)
        self.src_buf = src_buf
        self.color = int(color)
        self.mask_buf = mask_buf
        self.mask_res = mask_res
        self.mask_area = Area.of(mask_area) if mask_area is not None else self.blend_area
        self.opa = int(opa)
        self.blend_mode = blend_mode
        if src_format is None:
            src_format = SRC_FILL if src_buf is None else SRC_RGB565
        self.src_format = src_format
        self.flags = int(flags)


def _fill_batch(ctx, dsc):
    b = BlendBatch()
    b.dst = ctx.buf.data_ptr()
    b.dst_frame_stride = ctx.buf.stride(0) if ctx.n_frames > 1 else ctx.buf.shape[1] * ctx.buf.shape[2]
    if ctx.screen_transp:
        b.dst_frame_stride = (ctx.buf.stride(0) // 3) if ctx.n_frames > 1 else ctx.buf.shape[1] * ctx.buf.shape[2]
        b.dst_argb8565 = 1
    b.buf_area, b.clip_area, b.blend_area = ctx.buf_area, ctx.clip_area, dsc.blend_area
    b.n_frames = ctx.n_frames
    b.src_format = dsc.src_format
    if dsc.src_format not in _SRC_ELEM_BYTES and dsc.src_format != SRC_FILL:
        b.src = _dev(dsc.src_buf, "src_buf").data_ptr()  # let the library reject the format
    elif dsc.src_format != SRC_FILL:
        s = _dev(dsc.src_buf, "src_buf")
        frame_dims = 3 if dsc.src_format == SRC_ARGB8565 else 2
        exp = (dsc.blend_area.h, dsc.blend_area.w) + ((3,) if frame_dims == 3 else ())
        if tuple(s.shape[-frame_dims:]) != exp:
            raise ValueError("src_buf frame shape %s, blend_area needs %s" % (tuple(s.shape[-frame_dims:]), exp))
        eb = _SRC_ELEM_BYTES[dsc.src_format]
        if (s.element_size() if frame_dims == 2 else 3) != eb:
            raise ValueError("src_buf element size does not match src_format")
        fsb = _frame_stride_bytes(s, ctx.n_frames, frame_dims)
        b.src = s.data_ptr()
        b.src_frame_stride = fsb // eb
    if dsc.mask_buf is not None:
        m = _dev(dsc.mask_buf, "mask_buf")
        if m.element_size() != 1 or tuple(m.shape[-2:]) != (dsc.mask_area.h, dsc.mask_area.w):
            raise ValueError("mask_buf must be uint8 of mask_area's shape")
        b.mask = m.data_ptr()
        b.mask_frame_stride = _frame_stride_bytes(m, ctx.n_frames, 2)
        b.mask_area = dsc.mask_area
    b.color = dsc.color
    b.mask_res = dsc.mask_res
    b.opa = dsc.opa
    b.blend_mode = dsc.blend_mode
    b.color_16_swap = ctx.color_16_swap
    b.flags = dsc.flags
    return b


def lv_draw_sw_blend(ctx, dsc, stream=None):
    """libraries/lvgl/src/draw/sw/lv_draw_sw_blend.c:95 applied to every frame of ctx.buf, in place."""
    b = _fill_batch(ctx, dsc)
    check(_lib.lib().lvgpu_blend_batch(C.byref(b), _stream(stream)))


class PreparedBlend:
    """One lv_draw_sw_blend (optionally + flush) call marshalled once: run() is a single call into the C ABI, for
    loops where the Python-side marshalling of lv_draw_sw_blend() would take longer than the kernel."""

    def __init__(self, ctx, dsc, rotation=None, swap_bytes=0, out=None):
        self._keep = (ctx, dsc, out)
        self.out = out
        if rotation is None:
            self._b = _fill_batch(ctx, dsc)
            self._fn = _lib.lib().lvgpu_blend_batch
        else:
            n, h, w = ctx.buf.shape
            if out is None:
                shape = (n, w, h) if rotation in (LV_DISP_ROT_90, LV_DISP_ROT_270) else (n, h, w)
                self.out = out = torch.empty(shape, dtype=ctx.buf.dtype, device=ctx.buf.device)
            bf = BlendFlushBatch()
            bf.blend = _fill_batch(ctx, dsc)
            bf.out, bf.out_frame_stride = out.data_ptr(), (out.stride(0) if n > 1 else h * w)
            bf.rotation, bf.swap_bytes = rotation, int(swap_bytes)
            self._b = bf
            self._fn = _lib.lib().lvgpu_blend_flush_batch
        self._ref = C.byref(self._b)

    def run(self, stream=None):
        check(self._fn(self._ref, _stream(stream)))


def lv_draw_sw_blend_noalloc(ctx, dsc, stream=None):
    """lv_draw_sw_blend through lvgpu_blend_batch_noalloc: same result, no device allocation inside the call."""
    b = _fill_batch(ctx, dsc)
    check(_lib.lib().lvgpu_blend_batch_noalloc(C.byref(b), _stream(stream)))


def flush_rotate(inp, rotation, swap_bytes=0, out=None, stream=None):
    """The copy/rotate of lvgl_port_flush_callback for a batch [n, H, W] of rendered frames.  Returns the panel
    images, [n, W, H] for 90/270 degrees and [n, H, W] otherwise."""
    _dev(inp, "inp")
    n, h, w = inp.shape
    if out is None:
        shape = (n, w, h) if rotation in (LV_DISP_ROT_90, LV_DISP_ROT_270) else (n, h, w)
        out = torch.empty(shape, dtype=inp.dtype, device=inp.device)
    f = FlushBatch()
    f.inp, f.in_frame_stride = inp.data_ptr(), (inp.stride(0) if n > 1 else h * w)
    f.out, f.out_frame_stride = out.data_ptr(), (out.stride(0) if n > 1 else h * w)
    f.width, f.height, f.rotation, f.swap_bytes, f.n_frames = w, h, rotation, int(swap_bytes), n
    check(_lib.lib().lvgpu_flush_rotate_batch(C.byref(f), _stream(stream)))
    return out


def blend_flush(ctx, dsc, rotation, swap_bytes=0, out=None, stream=None):
    """lv_draw_sw_blend followed by the flush of the whole buffer, fused into one pass where possible."""
    n, h, w = ctx.buf.shape
    if out is None:
        shape = (n, w, h) if rotation in (LV_DISP_ROT_90, LV_DISP_ROT_270) else (n, h, w)
        out = torch.empty(shape, dtype=ctx.buf.dtype, device=ctx.buf.device)
    bf = BlendFlushBatch()
    bf.blend = _fill_batch(ctx, dsc)
    bf.out, bf.out_frame_stride = out.data_ptr(), (out.stride(0) if n > 1 else h * w)
    bf.rotation, bf.swap_bytes = rotation, int(swap_bytes)
    check(_lib.lib().lvgpu_blend_flush_batch(C.byref(bf), _stream(stream)))
    return out


def _fill_list(ctx, dscs, clips):
    """An array of lvgpu_blend_batch_t for a display list; clips: one clip area per item (LVGL changes the clip
    area from object to object) or None for ctx.clip_area throughout."""
    arr = (BlendBatch * max(len(dscs), 1))()
    saved = ctx.clip_area
    try:
        for i, d in enumerate(dscs):
            if clips is not None:
                ctx.clip_area = Area.of(clips[i])
            arr[i] = _fill_batch(ctx, d)
    finally:
        ctx.clip_area = saved
    return arr


def blend_list(ctx, dscs, clips=None, stream=None):
    """The descriptors of a whole scene applied in order to every frame of ctx.buf as one launch; same result as
    calling lv_draw_sw_blend on each of them in turn."""
    arr = _fill_list(ctx, dscs, clips)
    check(_lib.lib().lvgpu_blend_list_batch(arr, len(dscs), _stream(stream)))


def blend_list_flush(ctx, dscs, rotation, swap_bytes=0, clips=None, out=None, stream=None):
    """blend_list followed by the flush of the whole buffer (panel images returned), in the same launch where the
    shapes allow it."""
    n, h, w = ctx.buf.shape[:3]
    if out is None:
        shape = (n, w, h) if rotation in (LV_DISP_ROT_90, LV_DISP_ROT_270) else (n, h, w)
        out = torch.empty(shape, dtype=ctx.buf.dtype, device=ctx.buf.device)
    arr = _fill_list(ctx, dscs, clips)
    geom = _fill_batch(ctx, BlendDsc(ctx.buf_area))
    check(_lib.lib().lvgpu_blend_list_flush_batch(C.byref(geom), arr, len(dscs), out.data_ptr(),
                                                   (out.stride(0) if n > 1 else h * w), rotation, int(swap_bytes),
                                                   _stream(stream)))
    return out


def flush_chunks(area, hor_res, ver_res, rotation, trans_size):
    """The transport chunks of the reference's flush callback for `area` (src/lv_port.c:466-566) as
    [(x_start, y_start, x_end, y_end, offset_px, n_px)]: panel rectangle (end exclusive) and the slice of the area's
    panel image that goes there.  Host arithmetic, no GPU needed."""
    a = Area.of(area)
    L = _lib.lib()
    n = check_count(L.lvgpu_flush_chunks(C.byref(a), hor_res, ver_res, rotation, trans_size, None, 0))
    arr = (_lib.FlushChunk * max(n, 1))()
    check_count(L.lvgpu_flush_chunks(C.byref(a), hor_res, ver_res, rotation, trans_size, arr, n))
    return [(c.x_start, c.y_start, c.x_end, c.y_end, c.offset_px, c.n_px) for c in arr[:n]]


def axs15231b_transactions(chunk, x_gap=0, y_gap=0, qspi=True, bits_per_pixel=16):
    """What the reference's panel driver hands to the panel IO layer for one transport chunk (a tuple from
    flush_chunks), src/esp_lcd_axs15231b.c:292-337, as [(is_color, cmd, param bytes, data_offset_px, data_len)]."""
    c = _lib.FlushChunk(*chunk, 0)
    out = (_lib.PanelTxn * 3)()
    n = check_count(_lib.lib().lvgpu_axs15231b_transactions(C.byref(c), x_gap, y_gap, 1 if qspi else 0, bits_per_pixel, out))
    return [(t.is_color, t.cmd, bytes(t.param[:t.param_len]), t.data_offset_px, t.data_len) for t in out[:n]]


def check_count(rc):
    if rc < 0:
        raise LvgpuError(rc, _lib.lib().lvgpu_last_error().decode())
    return rc


def image_convert(src, src_format, color_16_swap=1, chroma_key=None, recolor=0, recolor_opa=0, stream=None):
    """Colour and alpha planes of an image as lv_draw_sw_img_decoded prepares them (convert_cb + recolor);
    chroma_key: the stored LV_COLOR_CHROMA_KEY value for LV_IMG_CF_TRUE_COLOR_CHROMA_KEYED images, else None."""
    n = src.numel() // 3 if src_format == SRC_ARGB8565 else src.numel()
    color = torch.empty(n, dtype=torch.int16, device=src.device)
    alpha = torch.empty(n, dtype=torch.uint8, device=src.device)
    o = _lib.ImageOpts()
    o.chroma_keyed, o.chroma_key = (0, 0) if chroma_key is None else (1, int(chroma_key))
    o.recolor, o.recolor_opa = int(recolor), int(recolor_opa)
    check(_lib.lib().lvgpu_image_convert(src.data_ptr(), src_format, color.data_ptr(), alpha.data_ptr(), n, color_16_swap,
                                          C.byref(o), _stream(stream)))
    return color, alpha


class BlendList:
    """A display list marshalled once and replayed many times (the descriptors, their sources and masks and ctx.buf
    must stay alive and in place)."""

    def __init__(self, ctx, dscs, clips=None):
        self.ctx, self.n = ctx, len(dscs)
        self._keep = (list(dscs), clips)
        self.arr = _fill_list(ctx, dscs, clips)
        self.geom = _fill_batch(ctx, BlendDsc(ctx.buf_area))

    def run(self, stream=None):
        check(_lib.lib().lvgpu_blend_list_batch(self.arr, self.n, _stream(stream)))

    def compile(self, flush=False, rotation=0, swap_bytes=0):
        """Plan the list once and keep its tables on the device (lvgpu_list_create); replay() is then one launch."""
        self.release()
        h = C.c_void_p()
        check(_lib.lib().lvgpu_list_create(C.byref(self.geom), self.arr, self.n, 1 if flush else 0, rotation,
                                            int(swap_bytes), C.byref(h)))
        self._handle, self._flush, self._rotation = h, bool(flush), rotation
        return self

    def replay(self, buf=None, out=None, stream=None):
        """One launch of the compiled list over buf (default: the context's buffers); returns the panel images when the
        list was compiled with flush."""
        b = self.ctx.buf if buf is None else buf
        n, h, w = b.shape[:3]
        optr, ofs = None, 0
        if self._flush:
            if out is None:
                shape = (n, w, h) if self._rotation in (LV_DISP_ROT_90, LV_DISP_ROT_270) else (n, h, w)
                out = torch.empty(shape, dtype=b.dtype, device=b.device)
            optr, ofs = out.data_ptr(), (out.stride(0) if n > 1 else h * w)
        check(_lib.lib().lvgpu_list_replay(self._handle, b.data_ptr(), n, optr, ofs, _stream(stream)))
        return out

    def release(self):
        if getattr(self, "_handle", None):
            check(_lib.lib().lvgpu_list_destroy(self._handle))
            self._handle = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass

    def plan(self, flush=False, rotation=0):
        """What replaying the list costs (lvgpu_list_plan_t as a dict); nothing is drawn."""
        p = _lib.ListPlan()
        check(_lib.lib().lvgpu_blend_list_plan(C.byref(self.geom), self.arr, self.n, 1 if flush else 0, rotation, C.byref(p)))
        return p.as_dict()

    def run_flush(self, rotation, swap_bytes=0, out=None, stream=None):
        n, h, w = self.ctx.buf.shape[:3]
        if out is None:
            shape = (n, w, h) if rotation in (LV_DISP_ROT_90, LV_DISP_ROT_270) else (n, h, w)
            out = torch.empty(shape, dtype=self.ctx.buf.dtype, device=self.ctx.buf.device)
        check(_lib.lib().lvgpu_blend_list_flush_batch(C.byref(self.geom), self.arr, self.n, out.data_ptr(),
                                                       (out.stride(0) if n > 1 else h * w), rotation, int(swap_bytes),
                                                       _stream(stream)))
        return out


def color_from_rgb888(xrgb, color_16_swap=1, stream=None):
    """lv_color_hex over a uint32/int32 tensor of 0x00RRGGBB words."""
    out = torch.empty(xrgb.shape, dtype=torch.int16, device=xrgb.device)
    check(_lib.lib().lvgpu_color_from_rgb888(xrgb.data_ptr(), out.data_ptr(), xrgb.numel(), color_16_swap,
                                              _stream(stream)))
    return out


def color_to32(c16, color_16_swap=1, stream=None):
    out = torch.empty(c16.shape, dtype=torch.int32, device=c16.device)
    check(_lib.lib().lvgpu_color_to32(c16.data_ptr(), out.data_ptr(), c16.numel(), color_16_swap, _stream(stream)))
    return out


def image_split(src, src_format, color_16_swap=1, stream=None):
    n = src.numel() // 3 if src_format == SRC_ARGB8565 else src.numel()
    color = torch.empty(n, dtype=torch.int16, device=src.device)
    alpha = torch.empty(n, dtype=torch.uint8, device=src.device)
    check(_lib.lib().lvgpu_image_split(src.data_ptr(), src_format, color.data_ptr(), alpha.data_ptr(), n,
                                        color_16_swap, _stream(stream)))
    return color, alpha


def color_fill(buf, color, stream=None):
    check(_lib.lib().lvgpu_color_fill(buf.data_ptr(), int(color), buf.numel(), _stream(stream)))


def buffer_copy(dst, dst_area, src, src_area, stream=None):
    """lv_draw_sw_buffer_copy over batches dst [n, Hd, Wd], src [n, Hs, Ws]."""
    n = dst.shape[0]
    check(_lib.lib().lvgpu_buffer_copy_batch(dst.data_ptr(), dst.stride(0), dst.shape[2], Area.of(dst_area),
                                              src.data_ptr(), src.stride(0), src.shape[2], Area.of(src_area), n,
                                              _stream(stream)))


_SYNTH_DTYPE = {SYNTH_U16: torch.int16, SYNTH_MASK_UI: torch.uint8, SYNTH_MASK_UNI: torch.uint8,
                SYNTH_ARGB_UI: torch.int32, SYNTH_ARGB_UNI: torch.int32}


def synth(kind, seed, first_index, shape, device="cuda", stream=None, out=None):
    """Synthetic input of `shape`; `out`: a tensor of that shape and the kind's dtype to fill instead of a new one."""
    if out is None:
        out = torch.empty(shape, dtype=_SYNTH_DTYPE[kind], device=device)
    else:
        assert tuple(out.shape) == tuple(shape) and out.dtype == _SYNTH_DTYPE[kind] and out.is_contiguous()
    check(_lib.lib().lvgpu_synth(out.data_ptr(), kind, seed, first_index, out.numel(), _stream(stream)))
    return out


def checksum(t, first_index=0, stream=None):
    acc = torch.zeros(1, dtype=torch.int64, device=t.device)
    check(_lib.lib().lvgpu_checksum(t.data_ptr(), t.element_size(), first_index, t.numel(), acc.data_ptr(),
                                     _stream(stream)))
    return int(acc.item()) & 0xFFFFFFFFFFFFFFFF


class MaskedStream:
    """A HIP stream restricted to a set of compute units (measurement support: "1/8 GPU")."""

    def __init__(self, cu_bits, total_cus=256):
        words = (total_cus + 31) // 32
        arr = (C.c_uint32 * words)()
        for b in cu_bits:
            arr[b // 32] |= 1 << (b % 32)
        self.handle = C.c_void_p()
        check(_lib.lib().lvgpu_stream_create_cu_mask(C.byref(self.handle), arr, words))
        self.cu_bits = list(cu_bits)

    def synchronize(self):
        check(_lib.lib().lvgpu_stream_synchronize(self.handle))

    def census(self, n_blocks=4096):
        """Workgroups per XCD for a launch on this stream."""
        counts = torch.zeros(8, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        check(_lib.lib().lvgpu_xcc_census(counts.data_ptr(), n_blocks, self.handle))
        self.synchronize()
        return counts.cpu().tolist()

    def close(self):
        if self.handle:
            check(_lib.lib().lvgpu_stream_destroy(self.handle))
            self.handle = C.c_void_p()
