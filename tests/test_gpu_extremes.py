"""GPU parity at the ends of lv_coord_t's usable range (LV_COORD_MAX = 8191 with 16-bit coordinates,
libraries/lvgl/src/misc/lv_area.h:31-35): buffers 8191 pixels wide or tall, buffer areas that do not start at 0 and
lie at negative coordinates, blend areas that stick out of the coordinate range's far ends, one-pixel buffers."""
import numpy as np
import pytest

from util import rand_mask, to_dev, to_np

pytestmark = pytest.mark.gpu

LV_COORD_MAX = 8191


def _blend_case(gpu, oracle, rng, swap, buf_area, clip, ba, n_frames, use_src, use_mask, opa):
    W, H = buf_area[2] - buf_area[0] + 1, buf_area[3] - buf_area[1] + 1
    bw, bh = ba[2] - ba[0] + 1, ba[3] - ba[1] + 1
    dst = rng.integers(0, 65536, (n_frames, H, W), dtype=np.uint16)
    src = rng.integers(0, 65536, (n_frames, bh, bw), dtype=np.uint16) if use_src else None
    mask = rand_mask(rng, (n_frames, bh, bw)) if use_mask else None
    color = int(rng.integers(0, 65536))
    exp = dst.copy()
    for f in range(n_frames):
        oracle.blend(exp[f], buf_area, clip, ba, src=None if src is None else src[f], color=color,
                     mask=None if mask is None else mask[f], opa=opa, swap=swap)
    d = to_dev(dst)
    ctx = gpu.DrawCtx(d, buf_area, clip, color_16_swap=swap)
    gpu.lv_draw_sw_blend(ctx, gpu.BlendDsc(ba, src_buf=None if src is None else to_dev(src), color=color,
                                           mask_buf=None if mask is None else to_dev(mask), opa=opa))
    assert np.array_equal(to_np(d, np.uint16), exp), (buf_area, clip, ba, use_src, use_mask, opa, gpu.last_kernel())


@pytest.mark.parametrize("swap", [1, 0])
@pytest.mark.parametrize("size", [(LV_COORD_MAX, 3), (3, LV_COORD_MAX), (LV_COORD_MAX + 1, 2), (1, 1), (1, 600), (600, 1)])
def test_blend_on_buffers_at_the_coordinate_limit(gpu, oracle, swap, size):
    W, H = size
    rng = np.random.default_rng(W * 3 + H + swap)
    full = (0, 0, W - 1, H - 1)
    for use_src, use_mask, opa in ((False, False, 255), (False, False, 137), (False, True, 255), (True, False, 255),
                                   (True, True, 200), (True, False, 90)):
        _blend_case(gpu, oracle, rng, swap, full, full, full, 2, use_src, use_mask, opa)       # the whole buffer
    if W > 8 and H > 1:
        part = (5, 1, W - 4, H - 1)                                                              # ragged rows
        _blend_case(gpu, oracle, rng, swap, full, full, part, 2, True, True, 255)
        _blend_case(gpu, oracle, rng, swap, full, (7, 0, W - 2, H - 2), part, 2, False, True, 180)


@pytest.mark.parametrize("swap", [1, 0])
def test_blend_with_areas_far_from_the_origin(gpu, oracle, swap):
    """The buffer is a window somewhere in a large coordinate space (LVGL renders invalidated areas into a small draw
    buffer: buf_area = that area, lv_draw_sw_blend.c:117); blend areas start left of / above it at negative
    coordinates or run past it to the far end of the range."""
    rng = np.random.default_rng(17 + swap)
    for buf_area in ((-8192, -8192, -8192 + 99, -8192 + 9), (8191 - 129, 8191 - 6, 8191, 8191), (-37, 4000, 120, 4012),
                     (3000, -5, 3100, 3)):
        bx1, by1, bx2, by2 = buf_area
        for _ in range(12):
            x1 = int(rng.integers(bx1 - 40, bx2))
            y1 = int(rng.integers(by1 - 4, by2))
            x2 = int(rng.integers(max(x1, bx1), bx2 + 40))
            y2 = int(rng.integers(max(y1, by1), by2 + 4))
            ba = (max(x1, -8192), max(y1, -8192), min(x2, 8191), min(y2, 8191))
            cx1, cy1 = int(rng.integers(bx1, bx2 + 1)), int(rng.integers(by1, by2 + 1))
            clip = (cx1, cy1, int(rng.integers(cx1, bx2 + 1)), int(rng.integers(cy1, by2 + 1)))
            _blend_case(gpu, oracle, rng, swap, buf_area, clip, ba, 2, bool(rng.integers(0, 2)), bool(rng.integers(0, 2)),
                        int(rng.choice([255, 254, 253, 180, 3])))


@pytest.mark.parametrize("rot", [0, 1, 2, 3])
@pytest.mark.parametrize("size", [(LV_COORD_MAX, 3), (3, LV_COORD_MAX), (LV_COORD_MAX + 1, 8), (8, LV_COORD_MAX + 1), (1, 1)])
def test_flush_of_frames_at_the_coordinate_limit(gpu, oracle, rot, size):
    W, H = size
    rng = np.random.default_rng(W + H * 5 + rot)
    frames = rng.integers(0, 65536, (2, H, W), dtype=np.uint16)
    out = gpu.flush_rotate(to_dev(frames), rot, swap_bytes=rot & 1)
    got = to_np(out, np.uint16).reshape(2, -1)
    for f in range(2):
        assert np.array_equal(got[f], oracle.flush_frame(frames[f], W, H, rot, rot & 1)), (size, rot, gpu.last_kernel())


def test_a_refused_allocation_does_not_fail_the_next_launch(gpu, oracle):
    """A runtime call that fails also stays behind as the thread's last error; the library takes it off, so that the
    launch check of the next call does not report an allocation refused before it."""
    import ctypes as C
    from vvc_amd import _lib
    p = C.c_void_p()
    assert _lib.lib().lvgpu_malloc(C.byref(p), 1 << 50) == -2          # LVGPU_ERR_HIP: a petabyte
    assert b"memory" in _lib.lib().lvgpu_last_error().lower()
    frames = np.arange(3 * 64 * 64, dtype=np.uint16).reshape(3, 64, 64)
    out = gpu.flush_rotate(to_dev(frames), 1)                            # raises if the launch check saw the old error
    got = to_np(out, np.uint16).reshape(3, -1)
    assert all(np.array_equal(got[f], oracle.flush_frame(frames[f], 64, 64, 1, 0)) for f in range(3))
    assert gpu.last_kernel() == "k_tile"
