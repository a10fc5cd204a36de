"""GPU: lvgpu_blend_batch_noalloc -- lvgpu_blend_batch without any device allocation inside the call; the per-frame
scan indices of the translucent unmasked fill live in a fixed array inside the library's device image.  Checked
bit for bit against lvgpu_blend_batch (and the CPU oracle) on two calls in a row, the second with a smaller input whose
frames need other indices than the first call left behind in the shared array."""
import ctypes as C

import numpy as np
import pytest
import torch

from util import rand_mask, to_dev, to_np

pytestmark = pytest.mark.gpu


def _frames(rng, n, H, W):
    """Destinations that exercise the first-pixel cache: black up to a different point in every frame, all-black
    frames, black after the first non-black pixel."""
    dst = rng.integers(1, 65536, (n, H, W), dtype=np.uint16)
    flat = dst.reshape(n, -1)
    for f in range(n):
        k = f % 4
        if k == 0:
            flat[f, : int(rng.integers(1, H * W))] = 0      # leading black of a random length
        elif k == 1:
            flat[f, :] = 0                                   # all black: the index stays "none"
        elif k == 2:
            flat[f, int(rng.integers(1, H * W)):] = 0        # black only after non-black
        else:
            flat[f, rng.random(H * W) < 0.5] = 0             # scattered black
    return dst


def _counters():
    from vvc_amd import _lib
    c = _lib.DebugCounters()
    _lib.lib().lvgpu_debug_counters(C.byref(c))
    return int(c.alloc_calls), int(c.first_nonblack_launches)


def _both(gpu, dst, area, clip, swap, prepass=None, **dsc_kw):
    """The same blend through lvgpu_blend_batch and through lvgpu_blend_batch_noalloc.  prepass: whether the call is
    expected to run the scan-order pre-pass, i.e. to need its scratch (None = do not check)."""
    d_ref, d_new = to_dev(dst), to_dev(dst)
    H, W = dst.shape[1:]
    buf_area = (0, 0, W - 1, H - 1)
    dsc = gpu.BlendDsc(area, **dsc_kw)
    a0, p0 = _counters()
    gpu.lv_draw_sw_blend(gpu.DrawCtx(d_ref, buf_area, clip, color_16_swap=swap), dsc)
    k_ref = gpu.last_kernel()
    a1, p1 = _counters()
    gpu.lv_draw_sw_blend_noalloc(gpu.DrawCtx(d_new, buf_area, clip, color_16_swap=swap), dsc)
    a2, p2 = _counters()
    assert gpu.last_kernel() == k_ref
    assert a2 == a1, "lvgpu_blend_batch_noalloc allocated"
    assert p2 - p1 == p1 - p0, "both entry points run the pre-pass for the same calls"
    if prepass is not None:
        # the control: the allocating entry point allocates exactly when the pre-pass runs, the other one never
        assert (p1 - p0, a1 - a0) == ((1, 1) if prepass else (0, 0))
    torch.cuda.synchronize()
    return to_np(d_ref, np.uint16), to_np(d_new, np.uint16)


def _oracle(oracle, dst, area, clip, color, opa, swap):
    exp = dst.copy()
    H, W = dst.shape[1:]
    for f in range(dst.shape[0]):
        oracle.blend(exp[f], (0, 0, W - 1, H - 1), clip, area, color=color, opa=opa, swap=swap)
    return exp


@pytest.mark.parametrize("swap", [1, 0])
def test_two_calls_in_a_row_second_smaller(gpu, oracle, swap):
    rng = np.random.default_rng(41 + swap)
    # colour / opacity pairs for which lv_color_mix and the premultiplied formula disagree on black, so the indices
    # matter (about one pair in twenty; every colour at opa 252, which the reference's uint8 arithmetic wraps to 0)
    color, opa = (0xD67F, 137) if swap else (0x2EDE, 103)
    W, H, n = 480, 320, 48
    area = (0, 0, W - 1, H - 1)
    dst = _frames(rng, n, H, W)
    dst[3].reshape(-1)[:1000] = 0        # frame 3: black, then non-black, then black again
    dst[3].reshape(-1)[2000:2100] = 0
    ref, new = _both(gpu, dst, area, area, swap, prepass=True, color=color, opa=opa)
    assert gpu.last_kernel() == "k_blend_linear"
    assert np.array_equal(new, ref)
    exp = _oracle(oracle, dst, area, area, color, opa, swap)
    assert np.array_equal(new, exp)
    # the indices matter for this colour/opacity: black before the first non-black pixel and black after it differ
    assert new[3].reshape(-1)[0] != new[3].reshape(-1)[2050]
    # second call: fewer frames, another size, a clipped and misaligned area; frame i now needs another index than
    # frame i of the first call left in the array (all-black frames need "none")
    W2, H2, n2 = 72, 40, 5
    area2, clip2 = (3, 2, 66, 37), (5, 0, 70, 33)
    dst2 = _frames(rng, n2, H2, W2)
    dst2[0] = 0
    dst2[2, :, :] = 0
    dst2[2, 20, 30] = 7
    ref2, new2 = _both(gpu, dst2, area2, clip2, swap, prepass=True, color=0xABCD, opa=252)
    assert gpu.last_kernel() == "k_blend_rows"
    assert np.array_equal(new2, ref2)
    assert np.array_equal(new2, _oracle(oracle, dst2, area2, clip2, 0xABCD, 252, swap))


def test_every_other_kind_of_blend_is_unchanged(gpu):
    rng = np.random.default_rng(43)
    W, H, n = 200, 96, 7
    area = (0, 0, W - 1, H - 1)
    dst = _frames(rng, n, H, W)
    src = to_dev(rng.integers(0, 1 << 32, (n, H, W), dtype=np.uint32))
    s16 = to_dev(rng.integers(0, 65536, (H, W), dtype=np.uint16))
    mask = to_dev(rand_mask(rng, (n, H, W)))
    for kw in (dict(src_buf=src, src_format=gpu.SRC_ARGB8888), dict(color=0x7BEF, mask_buf=mask, opa=200),
               dict(src_buf=s16, opa=90), dict(color=0xFFFF, opa=255), dict(src_buf=s16, mask_buf=mask, opa=255, blend_mode=1)):
        ref, new = _both(gpu, dst, area, area, 1, prepass=False, **kw)
        assert np.array_equal(new, ref), kw
    ref, new = _both(gpu, dst, (9, 7, 150, 80), (0, 0, 120, 90), 1, color=0xD67F, opa=137)
    assert np.array_equal(new, ref)


def test_more_frames_than_the_fixed_array_is_an_error(gpu):
    from vvc_amd import _lib
    d = torch.zeros((1, 8, 8), dtype=torch.int16, device="cuda")
    b = gpu._fill_batch(gpu.DrawCtx(d), gpu.BlendDsc((0, 0, 7, 7), color=1, opa=128))
    b.n_frames = 65537
    rc = _lib.lib().lvgpu_blend_batch_noalloc(C.byref(b), None)
    assert rc == -1
    assert b"at most 65536 frames" in _lib.lib().lvgpu_last_error()
    b.n_frames = 1
    assert _lib.lib().lvgpu_blend_batch_noalloc(C.byref(b), None) == 0
    torch.cuda.synchronize()


def test_under_stream_capture(gpu, oracle):
    """The call for which the entry point exists: captured into a HIP graph (where hipMalloc is not allowed) and
    replayed.  A translucent fill that needs the pre-pass, so the captured work is memset + pre-pass + blend."""
    rng = np.random.default_rng(47)
    W, H, n = 200, 96, 6
    area = (0, 0, W - 1, H - 1)
    dst = _frames(rng, n, H, W)
    exp = _oracle(oracle, dst, area, area, 0xD67F, 137, 1)
    d = to_dev(dst)
    ctx, dsc = gpu.DrawCtx(d), gpu.BlendDsc(area, color=0xD67F, opa=137)
    gpu.lv_draw_sw_blend_noalloc(ctx, dsc)                   # warm-up outside the capture: modules loaded, scratch looked up
    torch.cuda.synchronize()
    d.copy_(to_dev(dst))
    a0, _ = _counters()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        gpu.lv_draw_sw_blend_noalloc(ctx, dsc)               # on the capturing stream (torch's current stream)
    assert _counters()[0] == a0
    torch.cuda.synchronize()
    assert np.array_equal(to_np(d, np.uint16), dst), "capturing must not run anything"
    g.replay()
    torch.cuda.synchronize()
    assert np.array_equal(to_np(d, np.uint16), exp)
    # a second replay over other frames: the pre-pass is part of the graph and finds the new indices
    dst2 = _frames(rng, n, H, W)
    d.copy_(to_dev(dst2))
    g.replay()
    torch.cuda.synchronize()
    assert np.array_equal(to_np(d, np.uint16), _oracle(oracle, dst2, area, area, 0xD67F, 137, 1))
