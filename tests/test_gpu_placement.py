"""GPU: buffers that a host spreads over the three 96-GB regions of the device memory (vvc_amd/placement.py) give the
same results as buffers the allocator packs together, the ballast that pushes them apart is given back, and a device
too full for it is left alone.  What the spread buys in time is measured by bench.py, not asserted here."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

W, H, N = 480, 320, 64          # 19.7 MB of draw buffers: above the caching allocator's pooled sizes, so each is a fresh allocation


def _inputs(gpu):
    dst = gpu.synth(gpu.SYNTH_U16, 11, 0, (N, H, W))
    src = gpu.synth(gpu.SYNTH_ARGB_UI, 12, 0, (N, H, W))
    out = torch.empty((N, W, H), dtype=torch.int16, device="cuda")
    return dst, src, out


def _run(gpu, dst, src, out):
    ctx = gpu.DrawCtx(dst, color_16_swap=1)
    dsc = gpu.BlendDsc((0, 0, W - 1, H - 1), src_buf=src, opa=255, src_format=gpu.SRC_ARGB8888)
    gpu.PreparedBlend(ctx, dsc, 1, 0, out).run()
    torch.cuda.synchronize()
    return dst.cpu().numpy().copy(), out.cpu().numpy().copy()


def test_spread_buffers_same_results_and_ballast_returned(gpu):
    from vvc_amd.placement import RegionSpacer, REGION_BYTES
    torch.cuda.empty_cache()
    free0, total = torch.cuda.mem_get_info()
    if total < 3 * REGION_BYTES - (8 << 30) or total - free0 > (32 << 30):
        pytest.skip("needs a 288-GB device with little in use")
    sp = RegionSpacer()
    dst = gpu.synth(gpu.SYNTH_U16, 11, 0, (N, H, W))
    assert sp.next_region()
    src = gpu.synth(gpu.SYNTH_ARGB_UI, 12, 0, (N, H, W))
    assert sp.next_region()
    out = torch.empty((N, W, H), dtype=torch.int16, device="cuda")
    held = total - torch.cuda.mem_get_info()[0]
    assert held > 2 * REGION_BYTES                      # the panel images were allocated with two regions in use
    sp.close()
    assert sp.regions == [0, 1, 2] and sp.skipped is None and "0/1/2" in sp.describe()
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 < (1 << 30)                    # the ballast is back; only the three small buffers remain
    # the three lie far apart (the virtual addresses follow the allocation order, here: by more than 80 GB each)
    p = [t.data_ptr() for t in (dst, src, out)]
    assert abs(p[1] - p[0]) > (80 << 30) and abs(p[2] - p[1]) > (80 << 30)
    got = _run(gpu, dst, src, out)
    exp = _run(gpu, *_inputs(gpu))
    assert np.array_equal(got[0], exp[0]) and np.array_equal(got[1], exp[1])


def test_full_device_is_left_alone(gpu):
    from vvc_amd.placement import RegionSpacer
    free0 = torch.cuda.mem_get_info()[0]
    sp = RegionSpacer(keep_free=1 << 50)                # as if everything free were spoken for
    assert sp.next_region() is False and sp.skipped and len(sp.regions) == 2 and sp.regions[0] == sp.regions[1]
    sp.close()
    assert torch.cuda.mem_get_info()[0] > free0 - (64 << 20)   # nothing taken (close also empties the allocator's cache)
    assert "too full" in sp.describe()
