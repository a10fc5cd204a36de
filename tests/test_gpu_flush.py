"""GPU parity: the flush copy/rotate and the fused blend+flush against the CPU oracle, bit-exact."""
import numpy as np
import pytest

from util import rand_mask, to_dev, to_np

pytestmark = pytest.mark.gpu


def quarter_turn_kernel(W, H, fused=False):
    """Heights that are multiples of 64 pixels (and low ones of 32): one tile per wavefront.  Otherwise kernels that write long panel-row runs: the
    any-size kernel (workgroups sized to the height) up to 256 rows, and for the rotation alone also up to 384 and from
    1281 to 1536 rows; otherwise the column kernel (a workgroup of up to sixteen wavefronts per column of tiles).  Sizes
    off the 8-pixel grid: any-size."""
    if W % 8 or H % 8 or (H % 32 and H <= 256):
        return "k_tile_any"
    if H % 32 and not fused and (H <= 384 or 1280 < H <= 1536):
        return "k_tile_any"
    return "k_tile_col" if H % 32 or (H % 64 and H > 128) else "k_tile"


@pytest.mark.parametrize("rot", [0, 1, 2, 3])
@pytest.mark.parametrize("shape", [(480, 320), (320, 480), (100, 37), (64, 64), (72, 8), (8, 136), (1, 1), (7, 3),
                                   (96, 200), (64, 40), (200, 344), (136, 72), (8, 8), (240, 135), (135, 240), (65, 129),
                                   # any-size kernel with 5, 6, 4 + 4 and 6 + 6 + 6 wavefronts; the column kernel with 10, 14,
                                   # 16 and 16 + 16 (nine of them idle)
                                   (482, 320), (481, 321), (40, 328), (33, 466), (24, 600), (16, 1080), (8, 840), (16, 1000),
                                   (8, 1544), (10, 1000), (21, 601), (8, 1200), (8, 1304)])
@pytest.mark.parametrize("swap_bytes", [0, 1])
def test_flush_rotate(gpu, oracle, rot, shape, swap_bytes):
    W, H = shape
    N = 3
    rng = np.random.default_rng(W * 7 + H + rot)
    frames = rng.integers(0, 65536, (N, H, W), dtype=np.uint16)
    out = gpu.flush_rotate(to_dev(frames), rot, swap_bytes=swap_bytes)
    if rot in (1, 3):
        assert gpu.last_kernel() == quarter_turn_kernel(W, H)
    else:  # a copy / reversal only needs the frame to be a whole number of 8-pixel groups
        assert gpu.last_kernel() == ("k_blend_linear" if (W * H) % 8 == 0 else "k_flush_linear_any")
    got = to_np(out, np.uint16).reshape(N, -1)
    for f in range(N):
        exp = oracle.flush_frame(frames[f], W, H, rot, swap_bytes)
        assert np.array_equal(got[f], exp), "frame %d" % f
        if swap_bytes == 0:
            # and the reference's own transport chunks, concatenated, are that same image
            chunks = oracle.flush((0, 0, W - 1, H - 1), frames[f], rot, max(W * H // 10, max(W, H)), W, H)
            cat = np.concatenate([c[4] for c in chunks])
            assert np.array_equal(cat, exp)


@pytest.mark.parametrize("rot", [0, 1, 2, 3])
@pytest.mark.parametrize("swap", [1, 0])
@pytest.mark.parametrize("fmt", [0, 1, 2, 4])
def test_blend_flush_fused(gpu, oracle, rot, swap, fmt):
    """Config 3: blend of the whole 480x320 buffer + rotation in one kernel; dst is updated in place as well."""
    W, H, N = 480, 320, 3
    rng = np.random.default_rng(rot * 10 + swap + fmt * 100)
    area = (0, 0, W - 1, H - 1)
    for opa in (255, 253, 128):
        dst = rng.integers(0, 65536, (N, H, W), dtype=np.uint16)
        dst[0, : H // 3] = 0
        color = int(rng.integers(0, 65536))
        mask = None
        if fmt == 0:
            src = None
            mask = rand_mask(rng, (N, H, W)) if opa != 128 else None
        elif fmt == 1:
            src = rng.integers(0, 65536, (N, H, W), dtype=np.uint16)
            mask = rand_mask(rng, (N, H, W))
        elif fmt == 2:
            src = (rng.integers(0, 1 << 32, (N, H, W), dtype=np.uint32) & 0xFFFFFF) | (
                rand_mask(rng, (N, H, W)).astype(np.uint32) << 24)
        else:
            src = rng.integers(0, 256, (N, H, W, 3), dtype=np.uint8)
            src[..., 2] = rand_mask(rng, (N, H, W))
        exp_dst = dst.copy()
        for f in range(N):
            if fmt in (0, 1):
                oracle.blend(exp_dst[f], area, area, area, src=None if src is None else src[f], color=color,
                             mask=None if mask is None else mask[f], opa=opa, swap=swap)
            else:
                oracle.img_blend(exp_dst[f], area, area, area, src[f], fmt, opa, swap, W)
        d = to_dev(dst)
        ctx = gpu.DrawCtx(d, color_16_swap=swap)
        dsc = gpu.BlendDsc(area, src_buf=None if src is None else to_dev(src), color=color,
                           mask_buf=None if mask is None else to_dev(mask), opa=opa,
                           src_format=fmt if fmt else None)
        swap_bytes = 0 if swap else 1
        out = gpu.blend_flush(ctx, dsc, rot, swap_bytes=swap_bytes)
        assert "fused" in gpu.last_kernel()
        assert np.array_equal(to_np(d, np.uint16), exp_dst), "draw buffer, opa %d" % opa
        got = to_np(out, np.uint16).reshape(N, -1)
        for f in range(N):
            exp = oracle.flush_frame(exp_dst[f], W, H, rot, swap_bytes)
            assert np.array_equal(got[f], exp), "panel image, opa %d frame %d" % (opa, f)


@pytest.mark.parametrize("rot", [1, 3])
@pytest.mark.parametrize("shape", [(480, 328), (64, 40), (200, 8), (136, 216), (72, 264), (64, 520), (72, 1000), (64, 1080),
                                   (8, 1544), (8, 1272)])
def test_blend_flush_fused_strip_kernel(gpu, oracle, rot, shape):
    """Heights that are not a multiple of 32 px: fused blend + rotation through the column kernels."""
    W, H = shape
    N = 3
    rng = np.random.default_rng(W + H + rot)
    area = (0, 0, W - 1, H - 1)
    for fmt, opa in ((2, 255), (1, 200), (0, 128)):
        dst = rng.integers(0, 65536, (N, H, W), dtype=np.uint16)
        dst[0, : H // 3] = 0
        color = int(rng.integers(0, 65536))
        mask = None
        if fmt == 2:
            src = (rng.integers(0, 1 << 32, (N, H, W), dtype=np.uint32) & 0xFFFFFF) | (
                rand_mask(rng, (N, H, W)).astype(np.uint32) << 24)
        elif fmt == 1:
            src = rng.integers(0, 65536, (N, H, W), dtype=np.uint16)
            mask = rand_mask(rng, (N, H, W))
        else:
            src = None
        exp_dst = dst.copy()
        for f in range(N):
            if fmt == 2:
                oracle.img_blend(exp_dst[f], area, area, area, src[f], fmt, opa, 1, W)
            else:
                oracle.blend(exp_dst[f], area, area, area, src=None if src is None else src[f], color=color,
                             mask=None if mask is None else mask[f], opa=opa, swap=1)
        d = to_dev(dst)
        dsc = gpu.BlendDsc(area, src_buf=None if src is None else to_dev(src), color=color,
                           mask_buf=None if mask is None else to_dev(mask), opa=opa, src_format=fmt if fmt else None)
        out = gpu.blend_flush(gpu.DrawCtx(d), dsc, rot)
        assert gpu.last_kernel() == quarter_turn_kernel(W, H, fused=True) + "(fused)"
        assert np.array_equal(to_np(d, np.uint16), exp_dst)
        got = to_np(out, np.uint16).reshape(N, -1)
        for f in range(N):
            assert np.array_equal(got[f], oracle.flush_frame(exp_dst[f], W, H, rot, 0)), (fmt, f)


@pytest.mark.parametrize("rot", [1, 3])
@pytest.mark.parametrize("shape", [(200, 96), (504, 64), (512, 128), (8, 32), (136, 160), (320, 480), (520, 64)])
def test_blend_flush_fused_band_kernel(gpu, oracle, rot, shape):
    """Masked fills and maps at full opacity of frames up to 512 px wide (heights on the 32-px grid) take the band
    kernel: whole rows through LDS.  Both byte orders, opacities on both sides of the thresholds, a leading black region."""
    W, H = shape
    N = 3
    rng = np.random.default_rng(W * 3 + H + rot)
    area = (0, 0, W - 1, H - 1)
    for swap in (1, 0):
        for fmt, opa in ((1, 255), (1, 200), (0, 255), (0, 180)):
            dst = rng.integers(0, 65536, (N, H, W), dtype=np.uint16)
            dst[1].reshape(-1)[: (W * H) // 2] = 0
            color = int(rng.integers(0, 65536))
            src = rng.integers(0, 65536, (N, H, W), dtype=np.uint16) if fmt == 1 else None
            mask = rand_mask(rng, (N, H, W))
            exp = dst.copy()
            for f in range(N):
                oracle.blend(exp[f], area, area, area, src=None if src is None else src[f], color=color, mask=mask[f],
                             opa=opa, swap=swap)
            d = to_dev(dst)
            dsc = gpu.BlendDsc(area, src_buf=None if src is None else to_dev(src), color=color, mask_buf=to_dev(mask),
                               opa=opa)
            out = gpu.blend_flush(gpu.DrawCtx(d, color_16_swap=swap), dsc, rot, swap_bytes=1 - swap)
            # the band kernel takes the plain mix (opacity at or above the threshold); everything else stays with the tiles
            assert gpu.last_kernel() == ("k_band(fused)" if W <= 512 and opa == 255 else quarter_turn_kernel(W, H, fused=True) + "(fused)")
            assert np.array_equal(to_np(d, np.uint16), exp), (swap, fmt, opa)
            got = to_np(out, np.uint16).reshape(N, -1)
            for f in range(N):
                assert np.array_equal(got[f], oracle.flush_frame(exp[f], W, H, rot, 1 - swap)), (swap, fmt, opa, f)


def test_blend_flush_unfused_shapes(gpu, oracle):
    """Partial blend area or odd geometry: same result through blend + rotate launches."""
    rng = np.random.default_rng(5)
    for (W, H, ba) in [(100, 37, (0, 0, 99, 36)), (480, 320, (10, 10, 200, 100)), (64, 64, (70, 70, 80, 80))]:
        N = 2
        area = (0, 0, W - 1, H - 1)
        dst = rng.integers(0, 65536, (N, H, W), dtype=np.uint16)
        bw, bh = ba[2] - ba[0] + 1, ba[3] - ba[1] + 1
        src = rng.integers(0, 65536, (bh, bw), dtype=np.uint16)
        mask = rand_mask(rng, (bh, bw))
        exp_dst = dst.copy()
        for f in range(N):
            oracle.blend(exp_dst[f], area, area, ba, src=src, mask=mask, opa=200, swap=1)
        d = to_dev(dst)
        out = gpu.blend_flush(gpu.DrawCtx(d), gpu.BlendDsc(ba, src_buf=to_dev(src), mask_buf=to_dev(mask), opa=200), 1)
        assert np.array_equal(to_np(d, np.uint16), exp_dst)
        got = to_np(out, np.uint16).reshape(N, -1)
        for f in range(N):
            assert np.array_equal(got[f], oracle.flush_frame(exp_dst[f], W, H, 1, 0))


@pytest.mark.parametrize("swap", [1, 0])
def test_rotate_against_lvgl_own_sw_rotate(gpu, swap):
    """Whole frames rotated by LVGL's own drv.sw_rotate (golden, recorded from the real library): the GPU rotation
    with the rotation value lvgpu_rotation_for_lvgl_sw_rotate() names must give the same panel image."""
    import os
    from vvc_amd import _lib
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lvgl_sw_rotate_swap%d.npz" % swap))
    for LW, LH in ((480, 320), (320, 480), (200, 96)):
        logical = z["logical_%dx%d" % (LW, LH)]
        d = to_dev(np.repeat(logical[None], 3, axis=0))
        for lv_rot in (1, 2, 3):
            mine = _lib.lib().lvgpu_rotation_for_lvgl_sw_rotate(lv_rot)
            assert mine == {1: 3, 2: 2, 3: 1}[lv_rot]
            out = to_np(gpu.flush_rotate(d, mine), np.uint16).reshape(3, -1)
            lvgl = z["lv_rot%d_%dx%d" % (lv_rot, LW, LH)].ravel()
            for f in range(3):
                assert np.array_equal(out[f], lvgl), (LW, LH, lv_rot)


@pytest.mark.parametrize("rot", [1, 3])
@pytest.mark.parametrize("shape", [(100, 37), (135, 240), (65, 129), (7, 3), (1, 1), (466, 30), (9, 513), (64, 520),
                                   (481, 321), (33, 466), (172, 320), (20, 1021), (12, 700)])
def test_blend_flush_fused_any_size(gpu, oracle, rot, shape):
    """A quarter-turn flush of frames off the 8-pixel grid is still one pass with the blend (k_tile_any): every source
    kind, both byte orders, the translucent fill with its scan-order rule, frames whose stride leaves every frame at
    another 2-byte alignment."""
    W, H = shape
    N = 3
    rng = np.random.default_rng(W * 5 + H + rot)
    area = (0, 0, W - 1, H - 1)
    for swap in (1, 0):
        for fmt, opa in ((0, 255), (0, 137 if swap else 103), (1, 200), (1, 255), (2, 255), (4, 180)):
            dst = rng.integers(0, 65536, (N, H, W), dtype=np.uint16)
            dst[1].reshape(-1)[: (W * H) // 2] = 0           # leading black for the translucent fill
            color = (0xD67F if swap else 0x2EDE) if opa in (137, 103) else int(rng.integers(0, 65536))
            src = mask = None
            if fmt == 1:
                src = rng.integers(0, 65536, (N, H, W), dtype=np.uint16)
                mask = rand_mask(rng, (N, H, W))
            elif fmt == 2:
                src = rng.integers(0, 1 << 32, (N, H, W), dtype=np.uint32)
            elif fmt == 4:
                src = rng.integers(0, 256, (N, H, W, 3), dtype=np.uint8)
            elif opa == 255:
                mask = rand_mask(rng, (N, H, W))
            exp = dst.copy()
            for f in range(N):
                if fmt in (2, 4):
                    oracle.img_blend(exp[f], area, area, area, src[f], fmt, opa, swap, W)
                else:
                    oracle.blend(exp[f], area, area, area, src=None if src is None else src[f], color=color,
                                 mask=None if mask is None else mask[f], opa=opa, swap=swap)
            d = to_dev(dst)
            dsc = gpu.BlendDsc(area, src_buf=None if src is None else to_dev(src), color=color,
                               mask_buf=None if mask is None else to_dev(mask), opa=opa, src_format=fmt if fmt else None)
            out = gpu.blend_flush(gpu.DrawCtx(d, color_16_swap=swap), dsc, rot, swap_bytes=1 - swap)
            assert gpu.last_kernel() == quarter_turn_kernel(W, H, fused=True) + "(fused)"
            assert np.array_equal(to_np(d, np.uint16), exp), (swap, fmt, opa)
            got = to_np(out, np.uint16).reshape(N, -1)
            for f in range(N):
                assert np.array_equal(got[f], oracle.flush_frame(exp[f], W, H, rot, 1 - swap)), (swap, fmt, opa, f)


@pytest.mark.parametrize("rot", [0, 1, 2, 3])
@pytest.mark.parametrize("shape,fmt", [((480, 320), 2), ((480, 320), 1), ((480, 328), 2), ((200, 96), 1), ((135, 240), 2),
                                       ((466, 30), 4), ((64, 64), 0), ((3840, 24), 2), ((72, 264), 2), ((64, 600), 1), ((8, 1544), 2)])
def test_blend_flush_discard_dst(gpu, oracle, rot, shape, fmt):
    """LVGPU_BLEND_FLAG_DISCARD_DST (the reference's full_refresh configuration, src/lv_port.c:241: the draw buffer is
    dead once flushed): the panel image is what the call gives without the flag, and wherever blend and flush run as one
    pass the draw buffer keeps the pixels it had."""
    W, H = shape
    N = 3
    rng = np.random.default_rng(W + 7 * H + rot + fmt)
    area = (0, 0, W - 1, H - 1)
    for swap in (1, 0):
        dst = rng.integers(0, 65536, (N, H, W), dtype=np.uint16)
        dst[1].reshape(-1)[: (W * H) // 2] = 0
        color, opa = int(rng.integers(0, 65536)), (255 if fmt else 137)
        src = mask = None
        if fmt == 1:
            src = rng.integers(0, 65536, (N, H, W), dtype=np.uint16)
            mask = rand_mask(rng, (N, H, W))
        elif fmt == 2:
            src = rng.integers(0, 1 << 32, (N, H, W), dtype=np.uint32)
        elif fmt == 4:
            src = rng.integers(0, 256, (N, H, W, 3), dtype=np.uint8)
        exp = dst.copy()
        for f in range(N):
            if fmt in (2, 4):
                oracle.img_blend(exp[f], area, area, area, src[f], fmt, opa, swap, W)
            else:
                oracle.blend(exp[f], area, area, area, src=None if src is None else src[f], color=color,
                             mask=None if mask is None else mask[f], opa=opa, swap=swap)
        d = to_dev(dst)
        dsc = gpu.BlendDsc(area, src_buf=None if src is None else to_dev(src), color=color,
                           mask_buf=None if mask is None else to_dev(mask), opa=opa, src_format=fmt if fmt else None,
                           flags=gpu.FLAG_DISCARD_DST)
        out = gpu.blend_flush(gpu.DrawCtx(d, color_16_swap=swap), dsc, rot, swap_bytes=1 - swap)
        one_pass = "fused" in gpu.last_kernel()
        got = to_np(out, np.uint16).reshape(N, -1)
        for f in range(N):
            assert np.array_equal(got[f], oracle.flush_frame(exp[f], W, H, rot, 1 - swap)), (swap, f, gpu.last_kernel())
        after = to_np(d, np.uint16)
        assert np.array_equal(after, dst if one_pass else exp), (swap, gpu.last_kernel())
        assert one_pass or (W * H) % 8, "every whole-buffer blend on the 8-pixel grid is expected to run as one pass"


def test_discard_dst_is_for_the_fused_call_only(gpu):
    d = to_dev(np.zeros((1, 8, 8), np.uint16))
    dsc = gpu.BlendDsc((0, 0, 7, 7), color=1, flags=gpu.FLAG_DISCARD_DST)
    with pytest.raises(gpu.LvgpuError) as e:
        gpu.lv_draw_sw_blend(gpu.DrawCtx(d), dsc)
    assert e.value.code == -1 and "flags" in str(e.value)
    # a blend that does not cover the buffer runs as two passes: the flag is accepted and the buffer holds the blend
    d2 = to_dev(np.zeros((1, 16, 16), np.uint16))
    out = gpu.blend_flush(gpu.DrawCtx(d2), gpu.BlendDsc((0, 0, 7, 7), color=0x1234, flags=gpu.FLAG_DISCARD_DST), 1)
    h = to_np(d2, np.uint16)
    assert (h[0, :8, :8] == 0x1234).all() and (h[0, 8:] == 0).all()
    assert (to_np(out, np.uint16).reshape(16, 16)[:8, 8:] == 0x1234).all()


@pytest.mark.parametrize("rot", [1, 3, 2])
@pytest.mark.parametrize("shape,fmt", [((64, 64), 2), ((128, 64), 1), ((100, 37), 4), ((480, 328), 2), ((72, 40), 0)])
def test_blend_flush_without_antialiasing(gpu, oracle, rot, shape, fmt):
    """LVGPU_BLEND_FLAG_NO_AA through lvgpu_blend_flush_batch: one pass for 0 / 180 degrees (the contiguous kernel rounds
    the masks), blend + flush as two passes for the quarter turns (their kernels are compiled without the switch)."""
    W, H = shape
    N = 2
    rng = np.random.default_rng(W + H + rot + fmt)
    area = (0, 0, W - 1, H - 1)
    for swap in (1, 0):
        dst = rng.integers(0, 65536, (N, H, W), dtype=np.uint16)
        color = int(rng.integers(0, 65536))
        near = lambda shp: rng.integers(120, 137, shp, dtype=np.uint8)  # noqa: E731  around the threshold
        src = mask = None
        if fmt == 1:
            src, mask = rng.integers(0, 65536, (N, H, W), dtype=np.uint16), near((N, H, W))
        elif fmt == 2:
            src = rng.integers(0, 1 << 24, (N, H, W), dtype=np.uint32) | (near((N, H, W)).astype(np.uint32) << 24)
        elif fmt == 4:
            src = rng.integers(0, 256, (N, H, W, 3), dtype=np.uint8)
            src[..., 2] = near((N, H, W))
        else:
            mask = near((N, H, W))
        exp = dst.copy()
        for f in range(N):
            if fmt in (2, 4):
                oracle.img_blend(exp[f], area, area, area, np.ascontiguousarray(src[f]), fmt, 255, swap, W, antialiasing=0)
            else:
                oracle.blend(exp[f], area, area, area, src=None if src is None else src[f], color=color, mask=mask[f].copy(),
                             opa=255, swap=swap, antialiasing=0)
        d = to_dev(dst)
        dsc = gpu.BlendDsc(area, src_buf=None if src is None else to_dev(src), color=color,
                           mask_buf=None if mask is None else to_dev(mask), opa=255, src_format=fmt if fmt else None,
                           flags=gpu.FLAG_NO_AA)
        out = gpu.blend_flush(gpu.DrawCtx(d, color_16_swap=swap), dsc, rot, swap_bytes=1 - swap)
        assert ("fused" in gpu.last_kernel()) == (rot == 2 and (W * H) % 8 == 0)
        assert np.array_equal(to_np(d, np.uint16), exp), (swap, gpu.last_kernel())
        got = to_np(out, np.uint16).reshape(N, -1)
        for f in range(N):
            assert np.array_equal(got[f], oracle.flush_frame(exp[f], W, H, rot, 1 - swap)), (swap, f)


def test_band_kernel_without_its_residency_pad(gpu, oracle, monkeypatch):
    """k_band asks for dynamic LDS it never touches so that one workgroup has a CU to itself; on a device (or runtime)
    that allows a workgroup no more than the kernel's own 64 KB the pad is left out and the launch goes ahead."""
    import ctypes as C
    from vvc_amd import _lib
    W, H, N = 200, 96, 3
    rng = np.random.default_rng(77)
    area = (0, 0, W - 1, H - 1)
    dst = rng.integers(0, 65536, (N, H, W), dtype=np.uint16)
    src = rng.integers(0, 65536, (N, H, W), dtype=np.uint16)
    mask = rand_mask(rng, (N, H, W))
    exp = dst.copy()
    for f in range(N):
        oracle.blend(exp[f], area, area, area, src=src[f], mask=mask[f], opa=255, swap=1)
    pads = []
    for limit in (None, 65536):
        if limit:
            monkeypatch.setenv("LVGPU_MAX_LDS_PER_BLOCK", str(limit))
        d = to_dev(dst)
        out = gpu.blend_flush(gpu.DrawCtx(d), gpu.BlendDsc(area, src_buf=to_dev(src), mask_buf=to_dev(mask), opa=255), 1)
        assert gpu.last_kernel() == "k_band(fused)"
        c = _lib.DebugCounters()
        _lib.lib().lvgpu_debug_counters(C.byref(c))
        pads.append(int(c.band_pad_bytes))
        assert np.array_equal(to_np(d, np.uint16), exp)
        got = to_np(out, np.uint16).reshape(N, -1)
        for f in range(N):
            assert np.array_equal(got[f], oracle.flush_frame(exp[f], W, H, 1, 0))
    assert pads[1] == 0 and pads[0] in (0, 40 * 1024)   # 0 on a stream confined to some CUs or with LVGPU_NO_RESIDENCY_CAP


@pytest.mark.parametrize("limit", [65536, 40960, 8192])
def test_tall_frames_where_a_workgroup_gets_less_lds(gpu, oracle, monkeypatch, limit):
    """The column and any-size kernels size a workgroup to the frame height, 8 KB of LDS per wavefront and up to 128 KB;
    on a device (or runtime) that allows a workgroup less, they take fewer wavefronts per workgroup and more workgroups
    per column.  Rotation alone and fused, an aligned and an unaligned frame of 600 to 1000 rows."""
    monkeypatch.setenv("LVGPU_MAX_LDS_PER_BLOCK", str(limit))
    rng = np.random.default_rng(limit)
    for (W, H) in ((24, 600), (21, 601), (16, 1000)):
        N = 2
        area = (0, 0, W - 1, H - 1)
        frames = rng.integers(0, 65536, (N, H, W), dtype=np.uint16)
        for rot in (1, 3):
            got = to_np(gpu.flush_rotate(to_dev(frames), rot), np.uint16).reshape(N, -1)
            assert gpu.last_kernel() == quarter_turn_kernel(W, H)
            for f in range(N):
                assert np.array_equal(got[f], oracle.flush_frame(frames[f], W, H, rot, 0)), (W, H, rot)
            src = rng.integers(0, 1 << 32, (N, H, W), dtype=np.uint32)
            exp = frames.copy()
            for f in range(N):
                oracle.img_blend(exp[f], area, area, area, src[f], gpu.SRC_ARGB8888, 255, 1, W)
            d = to_dev(frames)
            out = gpu.blend_flush(gpu.DrawCtx(d), gpu.BlendDsc(area, src_buf=to_dev(src), opa=255, src_format=gpu.SRC_ARGB8888), rot)
            assert gpu.last_kernel() == quarter_turn_kernel(W, H, fused=True) + "(fused)"
            assert np.array_equal(to_np(d, np.uint16), exp)
            got = to_np(out, np.uint16).reshape(N, -1)
            for f in range(N):
                assert np.array_equal(got[f], oracle.flush_frame(exp[f], W, H, rot, 0)), (W, H, rot)
