"""GPU: buffers handed out by region of the device memory (vvc_amd/placement.py) give the same results as buffers the
allocator places, do not overlap, and the arena is given back.  What the placement buys in time is measured by
bench.py; here only that the timing sweep finds more than one region on a device that is mostly free."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

W, H, N = 480, 320, 64


def _run(gpu, dst, src, out):
    ctx = gpu.DrawCtx(dst, color_16_swap=1)
    dsc = gpu.BlendDsc((0, 0, W - 1, H - 1), src_buf=src, opa=255, src_format=gpu.SRC_ARGB8888)
    gpu.PreparedBlend(ctx, dsc, 1, 0, out).run()
    torch.cuda.synchronize()
    return dst.cpu().numpy().copy(), out.cpu().numpy().copy()


def test_arena_regions_same_results_and_memory_returned(gpu):
    from vvc_amd.placement import RegionArena, GB
    torch.cuda.empty_cache()
    free0, total = torch.cuda.mem_get_info()
    if free0 < 250 * GB:
        pytest.skip("needs a 288-GB device with little in use")
    _check_arena(gpu, RegionArena())              # all views of the arena die with that frame
    torch.cuda.empty_cache()
    assert torch.cuda.mem_get_info()[0] > free0 - 1 * GB


def _check_arena(gpu, arena):
    arena.calibrate(gpu)
    print(arena.describe())
    assert len(arena.labels) == arena.n_slots >= 25 and arena.labels[0] == 0
    assert len(set(arena.labels)) >= 2, arena.describe()      # the sweep saw launches inside and across a region
    bufs = [arena.empty((N, H, W), torch.int16, 0), arena.empty((N, H, W), torch.int32, 1),
            arena.empty((N, W, H), torch.int16, 2), arena.empty((N, H, W), torch.int16, 0)]
    spans = sorted((b.data_ptr(), b.data_ptr() + b.numel() * b.element_size()) for b in bufs)
    assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:]))                 # no two overlap
    assert all(b.data_ptr() % (2 << 20) == 0 for b in bufs)
    lo, hi = arena.buf.data_ptr(), arena.buf.data_ptr() + arena.buf.numel()
    assert all(lo <= a and b <= hi for a, b in spans)
    # each lies in the span of its region
    for b, r in zip(bufs, (0, 1, 2, 0)):
        off = b.data_ptr() - lo - arena.base
        assert arena.spans[r][0] <= off < arena.spans[r][1]
    with pytest.raises(RuntimeError):
        arena.empty((1 << 20, 1 << 10, 1 << 8), torch.uint8, 1)               # 256 TB
    dst, src, out = bufs[:3]
    gpu.synth(gpu.SYNTH_U16, 11, 0, (N, H, W), out=dst)
    gpu.synth(gpu.SYNTH_ARGB_UI, 12, 0, (N, H, W), out=src)
    got = _run(gpu, dst, src, out)
    exp = _run(gpu, gpu.synth(gpu.SYNTH_U16, 11, 0, (N, H, W)), gpu.synth(gpu.SYNTH_ARGB_UI, 12, 0, (N, H, W)),
               torch.empty((N, W, H), dtype=torch.int16, device="cuda"))
    assert np.array_equal(got[0], exp[0]) and np.array_equal(got[1], exp[1])


def test_a_device_with_too_little_free_memory_raises(gpu):
    from vvc_amd.placement import RegionArena
    with pytest.raises(RuntimeError, match="GB free"):
        RegionArena(min_bytes=1 << 50)
