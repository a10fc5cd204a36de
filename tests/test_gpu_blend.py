"""GPU parity: lvgpu_blend_batch (through the C ABI) against the CPU oracle, bit-exact.

Mirrors how the reference would test lv_draw_sw_blend: a draw context (buffer + buf_area + clip_area), a blend
descriptor, and the resulting buffer compared byte for byte.
"""
import numpy as np
import pytest

from util import OPAS, rand_mask, to_dev, to_np

pytestmark = pytest.mark.gpu


def _oracle_frames(O, dst, buf_area, clip, blend_area, src, color, mask, mask_res, mask_area, opa, mode, swap, fmt=1,
                   hor_res=480):
    out = dst.copy()
    for f in range(out.shape[0]):
        s = None if src is None else (src if src.ndim == (2 if fmt != 4 else 3) else src[f])
        m = None if mask is None else (mask if mask.ndim == 2 else mask[f])
        if fmt in (0, 1):
            O.blend(out[f], buf_area, clip, blend_area, src=s, color=color, mask=m, mask_res=mask_res,
                    mask_area=mask_area, opa=opa, blend_mode=mode, swap=swap)
        else:
            # image with its own alpha: the oracle restates lv_draw_sw_img_decoded (chunked split + blend)
            inter = O.Area()
            if not O.lib().lvref_area_intersect(__import__("ctypes").byref(inter), __import__("ctypes").byref(
                    O.Area.of(clip)), __import__("ctypes").byref(O.Area.of(blend_area))):
                continue
            O.img_blend(out[f], buf_area, inter.tup(), blend_area, np.ascontiguousarray(s), fmt, opa, swap, hor_res)
    return out


@pytest.mark.parametrize("swap", [1, 0])
def test_blend_random_rects(gpu, oracle, swap):
    """Clipped, misaligned rectangles; fill and map; every mask_res; opa across all thresholds (generic kernel)."""
    rng = np.random.default_rng(100 + swap)
    W, H, N = 67, 19, 3
    buf_area = (0, 0, W - 1, H - 1)
    for t in range(300):
        dst = rng.integers(0, 65536, (N, H, W), dtype=np.uint16)
        if t % 3 == 0:
            dst[rng.random(dst.shape) < 0.6] = 0
        if t % 7 == 0:
            dst[:, : H // 2] = 0
        x1, y1 = int(rng.integers(-3, W - 2)), int(rng.integers(-2, H - 1))
        x2, y2 = int(rng.integers(x1, W + 3)), int(rng.integers(y1, H + 2))
        ba = (x1, y1, x2, y2)
        cx1, cy1 = int(rng.integers(0, W)), int(rng.integers(0, H))
        clip = (cx1, cy1, int(rng.integers(cx1, W)), int(rng.integers(cy1, H)))
        bw, bh = x2 - x1 + 1, y2 - y1 + 1
        src = None
        if rng.random() < 0.5:
            src = rng.integers(0, 65536, (N, bh, bw) if rng.random() < 0.5 else (bh, bw), dtype=np.uint16)
        ma = (x1 - int(rng.integers(0, 3)), y1 - int(rng.integers(0, 3)), x2 + int(rng.integers(0, 3)),
              y2 + int(rng.integers(0, 3)))
        mask = None
        if rng.random() < 0.6:
            shp = (ma[3] - ma[1] + 1, ma[2] - ma[0] + 1)
            mask = rand_mask(rng, (N,) + shp if rng.random() < 0.5 else shp)
        opa = int(rng.choice(OPAS))
        mres = int(rng.choice([0, 1, 2, 2, 2, 3]))
        color = int(rng.integers(0, 65536))
        mode = 0
        exp = _oracle_frames(oracle, dst, buf_area, clip, ba, src, color, mask, mres, ma, opa, mode, swap)

        d = to_dev(dst)
        ctx = gpu.DrawCtx(d, buf_area, clip, color_16_swap=swap)
        dsc = gpu.BlendDsc(ba, src_buf=None if src is None else to_dev(src), color=color,
                           mask_buf=None if mask is None else to_dev(mask), mask_res=mres, mask_area=ma, opa=opa,
                           blend_mode=mode)
        gpu.lv_draw_sw_blend(ctx, dsc)
        got = to_np(d, np.uint16)
        assert np.array_equal(got, exp), "trial %d opa %d mres %d src %s mask %s" % (t, opa, mres, src is not None,
                                                                                      mask is not None)


@pytest.mark.parametrize("swap", [1, 0])
@pytest.mark.parametrize("shape", [(480, 320), (320, 480), (64, 8), (8, 8), (72, 40)])
def test_blend_full_frame_fast_path(gpu, oracle, swap, shape):
    """Full-frame blends take the vectorised kernel; all fill/map branches and opa thresholds."""
    W, H = shape
    N = 3
    rng = np.random.default_rng(7 + W + swap)
    area = (0, 0, W - 1, H - 1)
    cases = []
    for opa in OPAS:
        for has_src in (0, 1):
            for has_mask in (0, 1):
                cases.append((opa, has_src, has_mask))
    for opa, has_src, has_mask in cases:
        dst = rng.integers(0, 65536, (N, H, W), dtype=np.uint16)
        dst[0, : H // 2] = 0          # leading black
        dst[1] = 0                    # all black
        dst[2, H // 2:] = 0           # black after non-black
        src = rng.integers(0, 65536, (N, H, W), dtype=np.uint16) if has_src else None
        mask = rand_mask(rng, (N, H, W)) if has_mask else None
        color = int(rng.integers(0, 65536))
        exp = _oracle_frames(oracle, dst, area, area, area, src, color, mask, 2, area, opa, 0, swap)
        d = to_dev(dst)
        ctx = gpu.DrawCtx(d, color_16_swap=swap)
        dsc = gpu.BlendDsc(area, src_buf=None if src is None else to_dev(src), color=color,
                           mask_buf=None if mask is None else to_dev(mask), opa=opa)
        gpu.lv_draw_sw_blend(ctx, dsc)
        if opa > 2:
            assert gpu.last_kernel() == "k_blend_linear"
        got = to_np(d, np.uint16)
        assert np.array_equal(got, exp), "opa %d src %d mask %d" % (opa, has_src, has_mask)


@pytest.mark.parametrize("swap", [1, 0])
@pytest.mark.parametrize("fmt", [2, 3, 4])
def test_blend_image_with_alpha(gpu, oracle, swap, fmt):
    """ARGB8888 / RGBA8888 / ARGB8565 images: the colour/alpha split fused in front of map_normal (config 2)."""
    rng = np.random.default_rng(55 + fmt + swap)
    N = 2
    for (W, H, clip, coords) in [(480, 320, None, None), (64, 24, (3, 2, 50, 20), (1, 1, 60, 22)),
                                 (40, 16, (0, 0, 39, 15), (0, 0, 39, 15))]:
        area = (0, 0, W - 1, H - 1)
        clip = clip or area
        coords = coords or area
        iw, ih = coords[2] - coords[0] + 1, coords[3] - coords[1] + 1
        for opa in (255, 254, 253, 252, 128, 3, 2):
            dst = rng.integers(0, 65536, (N, H, W), dtype=np.uint16)
            if fmt == 4:
                src = rng.integers(0, 256, (N, ih, iw, 3), dtype=np.uint8)
                src[..., 2] = rand_mask(rng, (N, ih, iw))
            else:
                src = rng.integers(0, 1 << 32, (N, ih, iw), dtype=np.uint32)
                a = rand_mask(rng, (N, ih, iw)).astype(np.uint32)
                src = (src & 0xFFFFFF) | (a << 24)
            exp = _oracle_frames(oracle, dst, area, clip, coords, src, 0, None, 2, coords, opa, 0, swap, fmt=fmt,
                                 hor_res=W)
            d = to_dev(dst)
            ctx = gpu.DrawCtx(d, area, clip, color_16_swap=swap)
            dsc = gpu.BlendDsc(coords, src_buf=to_dev(src), opa=opa, src_format=fmt)
            gpu.lv_draw_sw_blend(ctx, dsc)
            got = to_np(d, np.uint16)
            assert np.array_equal(got, exp), "fmt %d opa %d %dx%d" % (fmt, opa, W, H)


def test_blend_noop_and_errors(gpu):
    import torch
    d = torch.zeros((2, 8, 8), dtype=torch.int16, device="cuda")
    ctx = gpu.DrawCtx(d)
    gpu.lv_draw_sw_blend(ctx, gpu.BlendDsc((20, 20, 30, 30), color=0xFFFF))          # outside the clip: no-op
    gpu.lv_draw_sw_blend(ctx, gpu.BlendDsc((0, 0, 7, 7), color=0xFFFF, opa=2))       # opa <= LV_OPA_MIN: no-op
    assert int(d.abs().sum()) == 0
    bad = gpu.DrawCtx(d, clip_area=(0, 0, 20, 20))                                   # clip outside the buffer
    with pytest.raises(gpu.LvgpuError):
        gpu.lv_draw_sw_blend(bad, gpu.BlendDsc((0, 0, 20, 20), color=1))
    with pytest.raises(gpu.LvgpuError):                                              # unknown source format
        gpu.lv_draw_sw_blend(ctx, gpu.BlendDsc((0, 0, 7, 7), src_buf=d, src_format=9))
    e = torch.zeros((0, 8, 8), dtype=torch.int16, device="cuda")                     # empty batch
    gpu.lv_draw_sw_blend(gpu.DrawCtx(e), gpu.BlendDsc((0, 0, 7, 7), color=1))


@pytest.mark.parametrize("swap", [1, 0])
@pytest.mark.parametrize("mode", [1, 2, 3, 4])
def test_blend_modes(gpu, oracle, swap, mode):
    """Additive / subtractive / multiply (fill_blended, map_blended) incl. the swapped build's green quirks;
    mode 4 (REPLACE) is 'unsupported' in the reference: a warning and no drawing."""
    rng = np.random.default_rng(900 + mode + 10 * swap)
    N = 2
    for (W, H, ba, clip) in [(96, 40, (0, 0, 95, 39), (0, 0, 95, 39)), (67, 19, (-2, 3, 60, 17), (1, 1, 64, 15))]:
        area = (0, 0, W - 1, H - 1)
        bw, bh = ba[2] - ba[0] + 1, ba[3] - ba[1] + 1
        for opa in (2, 3, 128, 252, 253, 255):
            for has_src in (0, 1):
                for has_mask in (0, 1):
                    dst = rng.integers(0, 65536, (N, H, W), dtype=np.uint16)
                    src = rng.integers(0, 65536, (N, bh, bw), dtype=np.uint16) if has_src else None
                    mask = rand_mask(rng, (N, bh, bw)) if has_mask else None
                    color = int(rng.integers(0, 65536))
                    exp = _oracle_frames(oracle, dst, area, clip, ba, src, color, mask, 2, ba, opa, mode, swap)
                    d = to_dev(dst)
                    ctx = gpu.DrawCtx(d, area, clip, color_16_swap=swap)
                    dsc = gpu.BlendDsc(ba, src_buf=None if src is None else to_dev(src), color=color,
                                       mask_buf=None if mask is None else to_dev(mask), opa=opa, blend_mode=mode)
                    gpu.lv_draw_sw_blend(ctx, dsc)
                    assert np.array_equal(to_np(d, np.uint16), exp), (W, opa, has_src, has_mask)


@pytest.mark.parametrize("kernel", ["k_blend_argb_rows", "k_blend_argb_dst"])
@pytest.mark.parametrize("swap", [1, 0])
def test_blend_argb8565_destination(gpu, oracle, swap, kernel, monkeypatch):
    """screen_transp: 3-byte destination pixels, fill_argb / map_argb / lv_color_mix_with_alpha, incl. the
    mask-row quirk of fill_argb's opaque branch and the blend modes through set_px_argb_blend.  Through the kernel that
    takes eight pixels per lane (areas narrower than 8 pixels, rows whose last group is moved back) and through the
    one-pixel-per-lane kernel it replaced (LVGPU_ARGB_KERNEL=px)."""
    if kernel == "k_blend_argb_dst":
        monkeypatch.setenv("LVGPU_ARGB_KERNEL", "px")
    rng = np.random.default_rng(321 + swap)
    N = 2
    W, H = 41, 13
    area = (0, 0, W - 1, H - 1)
    for t in range(250):
        dst = rng.integers(0, 256, (N, H, W, 3), dtype=np.uint8)
        dst[..., 2] = rand_mask(rng, (N, H, W))
        x1, y1 = int(rng.integers(-2, W - 2)), int(rng.integers(-2, H - 1))
        x2, y2 = int(rng.integers(x1, W + 2)), int(rng.integers(y1, H + 2))
        ba = (x1, y1, x2, y2)
        cx1, cy1 = int(rng.integers(0, W)), int(rng.integers(0, H))
        clip = (cx1, cy1, int(rng.integers(cx1, W)), int(rng.integers(cy1, H)))
        bw, bh = x2 - x1 + 1, y2 - y1 + 1
        src = rng.integers(0, 65536, (N, bh, bw), dtype=np.uint16) if rng.random() < 0.5 else None
        mask = rand_mask(rng, (N, bh, bw)) if rng.random() < 0.6 else None
        opa = int(rng.choice(OPAS))
        mode = int(rng.choice([0, 0, 1, 2, 3, 4]))
        color = int(rng.integers(0, 65536))
        exp = dst.copy()
        for f in range(N):
            oracle.blend(exp[f], area, clip, ba, src=None if src is None else src[f], color=color,
                         mask=None if mask is None else mask[f], opa=opa, blend_mode=mode, swap=swap, screen_transp=1)
        d = to_dev(dst)
        ctx = gpu.DrawCtx(d, area, clip, color_16_swap=swap, screen_transp=1)
        dsc = gpu.BlendDsc(ba, src_buf=None if src is None else to_dev(src), color=color,
                           mask_buf=None if mask is None else to_dev(mask), opa=opa, blend_mode=mode)
        gpu.lv_draw_sw_blend(ctx, dsc)
        assert np.array_equal(to_np(d, np.uint8), exp), (t, opa, mode, src is not None, mask is not None)
        inter = (max(x1, cx1), max(y1, cy1), min(x2, clip[2]), min(y2, clip[3]))
        if inter[0] <= inter[2] and inter[1] <= inter[3] and opa > 2:
            assert gpu.last_kernel() in (kernel, "k_blend_argb_linear")


@pytest.mark.parametrize("swap", [1, 0])
def test_blend_argb8565_destination_wide_areas(gpu, oracle, swap):
    """The eight-pixels-per-lane kernel on areas of every width from 1 to 40 and a few wide ones, at every byte
    alignment of the first pixel (3 * x), with sources and masks larger than the area (strides), image sources with
    their own alpha, and more frames than one workgroup column takes."""
    rng = np.random.default_rng(654 + swap)
    W, H = 230, 9
    area = (0, 0, W - 1, H - 1)
    widths = list(range(1, 41)) + [63, 64, 65, 201, 222]
    for t, w in enumerate(widths):
        N = 1 + t % 3
        x1 = int(rng.integers(0, W - w + 1))
        y1, y2 = int(rng.integers(0, 3)), int(rng.integers(3, H))
        clip = (x1, y1, x1 + w - 1, y2)
        ba = (x1 - int(rng.integers(0, 3)), y1 - int(rng.integers(0, 2)), x1 + w - 1 + int(rng.integers(0, 3)), y2 + int(rng.integers(0, 2)))
        bw, bh = ba[2] - ba[0] + 1, ba[3] - ba[1] + 1
        dst = rng.integers(0, 256, (N, H, W, 3), dtype=np.uint8)
        dst[..., 2] = rand_mask(rng, (N, H, W))
        kind = t % 4
        mask = rand_mask(rng, (N, bh, bw)) if t % 5 else None
        opa = int(rng.choice(OPAS))
        color = int(rng.integers(0, 65536))
        exp = dst.copy()
        if kind < 2:
            src = rng.integers(0, 65536, (N, bh, bw), dtype=np.uint16) if kind == 1 else None
            for f in range(N):
                oracle.blend(exp[f], area, clip, ba, src=None if src is None else src[f], color=color,
                             mask=None if mask is None else mask[f], opa=opa, swap=swap, screen_transp=1)
            dsc = gpu.BlendDsc(ba, src_buf=None if src is None else to_dev(src), color=color,
                               mask_buf=None if mask is None else to_dev(mask), opa=opa)
        else:
            fmt = gpu.SRC_ARGB8888 if kind == 2 else gpu.SRC_ARGB8565
            if fmt == gpu.SRC_ARGB8888:
                src = rng.integers(0, 1 << 32, (N, bh, bw), dtype=np.uint32)
                src = (src & 0xFFFFFF) | (rand_mask(rng, (N, bh, bw)).astype(np.uint32) << 24)
            else:
                src = rng.integers(0, 256, (N, bh, bw, 3), dtype=np.uint8)
                src[..., 2] = rand_mask(rng, (N, bh, bw))
            # lv_draw_sw_img_decoded hands the blend a colour plane and the image's alpha as its mask (lv_draw_sw_img.c:133-136)
            for f in range(N):
                c16, a8 = oracle.image_split(np.ascontiguousarray(src[f]), fmt, swap)
                oracle.blend(exp[f], area, clip, ba, src=c16.reshape(bh, bw), mask=a8.reshape(bh, bw), opa=opa, swap=swap,
                             screen_transp=1)
            dsc = gpu.BlendDsc(ba, src_buf=to_dev(src), opa=opa, src_format=fmt)
        d = to_dev(dst)
        gpu.lv_draw_sw_blend(gpu.DrawCtx(d, area, clip, color_16_swap=swap, screen_transp=1), dsc)
        assert np.array_equal(to_np(d, np.uint8), exp), (w, kind, opa, mask is not None)
        if opa > 2:
            assert gpu.last_kernel() == "k_blend_argb_rows"


@pytest.mark.parametrize("swap", [1, 0])
def test_blend_argb8565_destination_full_frames(gpu, oracle, swap):
    """The same semantics over whole, aligned frames: the vector kernel (16 pixels = 48 bytes per lane; frames whose
    pixel count is not a multiple of 16 take the per-pixel kernel)."""
    rng = np.random.default_rng(77 + swap)
    N = 3
    for (W, H) in [(64, 24), (480, 320), (40, 5), (48, 5), (1088, 3)]:
        area = (0, 0, W - 1, H - 1)
        for t in range(40 if W < 480 else 8):
            dst = rng.integers(0, 256, (N, H, W, 3), dtype=np.uint8)
            dst[..., 2] = rand_mask(rng, (N, H, W))
            src = rng.integers(0, 65536, (N, H, W), dtype=np.uint16) if rng.random() < 0.5 else None
            mask = rand_mask(rng, (N, H, W)) if rng.random() < 0.6 else None
            opa = int(rng.choice(OPAS))
            mode = int(rng.choice([0, 0, 1, 2, 3, 4]))
            color = int(rng.integers(0, 65536))
            exp = dst.copy()
            for f in range(N):
                oracle.blend(exp[f], area, area, area, src=None if src is None else src[f], color=color,
                             mask=None if mask is None else mask[f], opa=opa, blend_mode=mode, swap=swap, screen_transp=1)
            d = to_dev(dst)
            ctx = gpu.DrawCtx(d, area, area, color_16_swap=swap, screen_transp=1)
            dsc = gpu.BlendDsc(area, src_buf=None if src is None else to_dev(src), color=color,
                               mask_buf=None if mask is None else to_dev(mask), opa=opa, blend_mode=mode)
            gpu.lv_draw_sw_blend(ctx, dsc)
            assert np.array_equal(to_np(d, np.uint8), exp), (W, t, opa, mode, src is not None, mask is not None)
            if opa > 2:
                assert gpu.last_kernel() == ("k_blend_argb_linear" if (W * H) % 16 == 0 else "k_blend_argb_rows")


@pytest.mark.parametrize("swap", [1, 0])
def test_argb8565_destination_every_alpha_pair(gpu, oracle, swap, monkeypatch):
    """lv_color_mix_with_alpha (lv_color.h:530-577) divides fg_opa * 255 by the resulting opacity; the kernels take a
    reciprocal instead: every (destination alpha, foreground alpha) pair, through the three ARGB8565 kernels."""
    rng = np.random.default_rng(5 + swap)
    H = W = 256
    area = (0, 0, W - 1, H - 1)
    for opa in (255, 254, 253, 200, 3):
        dst = rng.integers(0, 256, (2, H, W, 3), dtype=np.uint8)
        dst[..., 2] = np.arange(256, dtype=np.uint8)[None, :, None]      # destination alpha = row
        mask = np.broadcast_to(np.arange(256, dtype=np.uint8)[None, None, :], (2, H, W)).copy()  # mask = column
        src = rng.integers(0, 65536, (2, H, W), dtype=np.uint16)
        for use_src in (False, True):
            color = int(rng.integers(0, 65536))
            for clip, kernel in ((area, "k_blend_argb_linear"), ((1, 0, W - 1, H - 1), "k_blend_argb_rows"),
                                 ((1, 0, W - 1, H - 1), "k_blend_argb_dst")):
                if kernel == "k_blend_argb_dst":
                    monkeypatch.setenv("LVGPU_ARGB_KERNEL", "px")
                else:
                    monkeypatch.delenv("LVGPU_ARGB_KERNEL", raising=False)
                exp = dst.copy()
                for f in range(2):
                    oracle.blend(exp[f], area, clip, area, src=src[f] if use_src else None, color=color, mask=mask[f], opa=opa,
                                 swap=swap, screen_transp=1)
                d = to_dev(dst)
                ctx = gpu.DrawCtx(d, area, clip, color_16_swap=swap, screen_transp=1)
                gpu.lv_draw_sw_blend(ctx, gpu.BlendDsc(area, src_buf=to_dev(src) if use_src else None, color=color,
                                                       mask_buf=to_dev(mask), opa=opa))
                assert gpu.last_kernel() == kernel
                assert np.array_equal(to_np(d, np.uint8), exp), (opa, use_src, kernel)


@pytest.mark.parametrize("swap", [1, 0])
def test_translucent_fill_many_frames_one_block_per_frame(gpu, oracle, swap):
    """With thousands of frames the first-non-black pre-pass runs one workgroup per frame over the whole area (with few
    frames: several segments per frame); frames whose first non-black pixel lies beyond the first 1024-pixel piece,
    all-black frames, and a clipped area whose rows are not contiguous."""
    rng = np.random.default_rng(91 + swap)
    color, opa = (0xD67F, 137) if swap else (0x2EDE, 103)   # the two formulas disagree on black for these
    N, H, W = 3000, 40, 72
    dst = rng.integers(1, 65536, (N, H, W), dtype=np.uint16)
    flat = dst.reshape(N, -1)
    for f in range(N):
        k = f % 5
        if k == 0:
            flat[f, : int(rng.integers(1, H * W))] = 0
        elif k == 1:
            flat[f, :] = 0
        elif k == 2:
            flat[f, : 1024 + int(rng.integers(0, 1500))] = 0   # the first piece holds nothing but black
        elif k == 3:
            flat[f, rng.random(H * W) < 0.7] = 0
    for clip in ((0, 0, W - 1, H - 1), (5, 3, 60, 37)):
        exp = dst.copy()
        buf_area = (0, 0, W - 1, H - 1)
        for f in range(N):
            oracle.blend(exp[f], buf_area, clip, buf_area, color=color, opa=opa, swap=swap)
        d = to_dev(dst)
        gpu.lv_draw_sw_blend(gpu.DrawCtx(d, buf_area, clip, color_16_swap=swap), gpu.BlendDsc(buf_area, color=color, opa=opa))
        assert gpu.last_kernel() == ("k_blend_linear" if clip[0] == 0 else "k_blend_rows")
        got = to_np(d, np.uint16)
        bad = np.flatnonzero((got != exp).reshape(N, -1).any(axis=1))
        assert bad.size == 0, "frames %s differ" % bad[:8]


@pytest.mark.parametrize("swap", [1, 0])
def test_blend_without_antialiasing(gpu, oracle, swap):
    """LVGPU_BLEND_FLAG_NO_AA = a display with anti-aliasing off: every mask value (and the alpha of a source with its
    own alpha) is rounded to 0 / 255 before the blend (lv_draw_sw_blend.c:150-156).  Against the oracle -- itself pinned
    on the real LVGL for this case (tests/golden/aa_off_swap*.npz) -- through the clipped-area kernel, the contiguous
    kernels, both destination formats, single calls and display lists; masks crowd the threshold (127, 128, 129)."""
    rng = np.random.default_rng(900 + swap)

    def mask_of(shape):
        m = rand_mask(rng, shape)
        sel = rng.random(shape)
        m[sel < 0.25] = rng.integers(126, 131, int((sel < 0.25).sum()), dtype=np.uint8)
        return m

    for t in range(120):
        transp = int(t % 3 == 2)
        W, H = ((67, 19), (64, 64), (96, 8))[t % 3 if not transp else t % 2]
        N = 3
        full = (0, 0, W - 1, H - 1)
        if t % 2:
            ba = full                                          # whole buffer: the contiguous kernels
        else:
            x1, y1 = int(rng.integers(0, W - 8)), int(rng.integers(0, H - 2))
            ba = (x1, y1, int(rng.integers(x1 + 1, W)), int(rng.integers(y1 + 1, H)))
        bw, bh = ba[2] - ba[0] + 1, ba[3] - ba[1] + 1
        fmt = int(rng.choice([0, 1, 2, 4]))
        opa = int(rng.choice([255, 254, 253, 200]))
        mode = int(rng.choice([0, 0, 1, 3])) if fmt in (0, 1) else 0
        color = int(rng.integers(0, 65536))
        src = mask = None
        if fmt == 1:
            src = rng.integers(0, 65536, (N, bh, bw), dtype=np.uint16)
        elif fmt == 2:
            src = (rng.integers(0, 1 << 24, (N, bh, bw), dtype=np.uint32)) | (mask_of((N, bh, bw)).astype(np.uint32) << 24)
        elif fmt == 4:
            src = rng.integers(0, 256, (N, bh, bw, 3), dtype=np.uint8)
            src[..., 2] = mask_of((N, bh, bw))
        if fmt in (0, 1):
            mask = mask_of((N, bh, bw))
        if transp and fmt in (2, 4):
            fmt, src, mask = 0, None, mask_of((N, bh, bw))     # the reference draws such images through convert_cb first
        dst = rng.integers(0, 256, (N, H, W, 3), dtype=np.uint8) if transp else rng.integers(0, 65536, (N, H, W), dtype=np.uint16)
        exp = dst.copy()
        for f in range(N):
            if fmt in (0, 1):
                oracle.blend(exp[f], full, full, ba, src=None if src is None else src[f], color=color,
                             mask=None if mask is None else mask[f].copy(), opa=opa, blend_mode=mode, swap=swap,
                             screen_transp=transp, antialiasing=0)
            else:
                oracle.img_blend(exp[f], full, full, ba, np.ascontiguousarray(src[f]), fmt, opa, swap, 480, antialiasing=0)
        d = to_dev(dst)
        m_dev = None if mask is None else to_dev(mask)
        ctx = gpu.DrawCtx(d, full, full, color_16_swap=swap, screen_transp=transp)
        dsc = gpu.BlendDsc(ba, src_buf=None if src is None else to_dev(src), color=color, mask_buf=m_dev, opa=opa,
                           blend_mode=mode, src_format=fmt if fmt else None, flags=gpu.FLAG_NO_AA)
        if t % 4 < 2:
            gpu.lv_draw_sw_blend(ctx, dsc)
        else:
            gpu.blend_list(ctx, [dsc])
        got = to_np(d, np.uint8 if transp else np.uint16).reshape(exp.shape)
        assert np.array_equal(got, exp), (t, W, H, ba, fmt, opa, mode, transp, gpu.last_kernel())
        if mask is not None:
            assert np.array_equal(to_np(m_dev, np.uint8).reshape(mask.shape), mask), "the caller's mask is left as it is"
