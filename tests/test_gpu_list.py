"""GPU parity: display lists (lvgpu_blend_list_batch / lvgpu_blend_list_flush_batch) against the CPU oracle applied
descriptor by descriptor, bit-exact.  A list stands for the sequence of lv_draw_sw_blend calls LVGL's producers issue
for one scene (per object part / per row)."""
import numpy as np
import pytest

from test_gpu_blend import _oracle_frames
from util import OPAS, rand_mask, to_dev, to_np

pytestmark = pytest.mark.gpu


def _rand_item(rng, W, H, N, shared):
    """One random descriptor: (kind fields for the oracle, kwargs for the GPU descriptor)."""
    r = rng.random()
    if r < 0.08:      # full-screen
        ba = (0, 0, W - 1, H - 1)
    elif r < 0.35:    # one row, as the row-by-row producers emit
        x1 = int(rng.integers(-3, W - 1))
        y = int(rng.integers(0, H))
        ba = (x1, y, int(rng.integers(x1, W + 3)), y)
    else:
        x1, y1 = int(rng.integers(-4, W - 1)), int(rng.integers(-4, H - 1))
        ba = (x1, y1, int(rng.integers(x1, min(W + 4, x1 + 150))), int(rng.integers(y1, min(H + 4, y1 + 100))))
    if rng.random() < 0.5:
        clip = (0, 0, W - 1, H - 1)
    else:
        cx1, cy1 = int(rng.integers(0, W)), int(rng.integers(0, H))
        clip = (cx1, cy1, int(rng.integers(cx1, W)), int(rng.integers(cy1, H)))
    bw, bh = ba[2] - ba[0] + 1, ba[3] - ba[1] + 1
    n = () if shared else (N,)
    kind = rng.choice(["fill", "map", "img"], p=[0.55, 0.3, 0.15])
    opa = int(rng.choice(OPAS + [255, 255, 255]))
    it = dict(ba=ba, clip=clip, opa=opa, color=int(rng.integers(0, 65536)), src=None, mask=None, mode=0, fmt=0,
              mask_res=2)
    if kind == "img":
        fmt = int(rng.choice([2, 3, 4]))
        it["fmt"] = fmt
        if fmt == 4:
            src = rng.integers(0, 256, n + (bh, bw, 3), dtype=np.uint8)
            src[..., 2] = rand_mask(rng, n + (bh, bw))
        else:
            src = rng.integers(0, 1 << 32, n + (bh, bw), dtype=np.uint32)
            src = (src & 0xFFFFFF) | (rand_mask(rng, n + (bh, bw)).astype(np.uint32) << 24)
        it["src"] = src
        return it
    if kind == "map":
        it["fmt"] = 1
        it["src"] = rng.integers(0, 65536, n + (bh, bw), dtype=np.uint16)
    it["mode"] = int(rng.choice([0, 0, 0, 0, 1, 2, 3, 4]))
    if rng.random() < 0.6:
        it["mask"] = rand_mask(rng, n + (bh, bw))
        it["mask_res"] = int(rng.choice([0, 1, 2, 2, 2, 3]))
    return it


def _apply_oracle(O, dst, area, items, swap, W):
    exp = dst
    for it in items:
        exp = _oracle_frames(O, exp, area, it["clip"], it["ba"], it["src"], it["color"], it["mask"], it["mask_res"],
                             it["ba"], it["opa"], it["mode"], swap, fmt=it["fmt"], hor_res=W)
    return exp


def _gpu_items(gpu, items):
    dscs, clips, keep = [], [], []
    for it in items:
        s = None if it["src"] is None else to_dev(it["src"])
        m = None if it["mask"] is None else to_dev(it["mask"])
        keep += [s, m]
        dscs.append(gpu.BlendDsc(it["ba"], src_buf=s, color=it["color"], mask_buf=m, mask_res=it["mask_res"],
                                 opa=it["opa"], blend_mode=it["mode"], src_format=it["fmt"]))
        clips.append(it["clip"])
    return dscs, clips, keep


@pytest.mark.parametrize("swap", [1, 0])
@pytest.mark.parametrize("W,H,N,shared", [(128, 96, 3, False), (480, 320, 2, True), (72, 50, 3, False), (67, 19, 2, False)])
def test_list_matches_sequential_oracle(gpu, oracle, swap, W, H, N, shared):
    """Random scenes of 60 descriptors; some frames carry black regions so that the scan-order cache of the
    translucent fill matters and such items have to be taken out of the tile pass in their place."""
    rng = np.random.default_rng(4242 + swap + W)
    area = (0, 0, W - 1, H - 1)
    for scene in range(4):
        dst = rng.integers(0, 65536, (N, H, W), dtype=np.uint16)
        dst[:, : H // 3, : W // 2] = 0
        items = [_rand_item(rng, W, H, N, shared) for _ in range(60)]
        if scene % 2 == 0:  # an opaque background first: tiles need not be loaded
            items[0] = dict(ba=area, clip=area, opa=255, color=0x1234, src=None, mask=None, mode=0, fmt=0, mask_res=2)
            items[1] = dict(ba=(0, 0, W // 2, H - 1), clip=area, opa=128, color=int(rng.integers(0, 65536)), src=None,
                            mask=None, mode=0, fmt=0, mask_res=2)
        exp = _apply_oracle(oracle, dst, area, items, swap, W)
        d = to_dev(dst)
        ctx = gpu.DrawCtx(d, area, area, color_16_swap=swap)
        dscs, clips, keep = _gpu_items(gpu, items)
        gpu.blend_list(ctx, dscs, clips)
        got = to_np(d, np.uint16)
        assert np.array_equal(got, exp), "scene %d: %d pixels differ" % (scene, int((got != exp).sum()))
        if W % 8 == 0:
            assert gpu.last_kernel() == "k_blend_list"


@pytest.mark.parametrize("swap", [1, 0])
@pytest.mark.parametrize("rot", [0, 1, 2, 3])
def test_list_flush(gpu, oracle, swap, rot):
    rng = np.random.default_rng(77 + rot)
    for (W, H, N) in [(128, 64, 2), (480, 320, 2), (200, 72, 3), (96, 100, 2), (50, 30, 2)]:
        area = (0, 0, W - 1, H - 1)
        dst = rng.integers(0, 65536, (N, H, W), dtype=np.uint16)
        items = [_rand_item(rng, W, H, N, False) for _ in range(25)]
        for it in items:  # keep the last segment a tile pass (no scan-order items): that is the fused launch
            if it["fmt"] == 0 and it["mask"] is None and it["mode"] == 0 and 2 < it["opa"] < 253:
                it["opa"] = 255
        exp = _apply_oracle(oracle, dst, area, items, swap, W)
        d = to_dev(dst)
        ctx = gpu.DrawCtx(d, area, area, color_16_swap=swap)
        dscs, clips, keep = _gpu_items(gpu, items)
        out = gpu.blend_list_flush(ctx, dscs, rot, swap_bytes=rot & 1, clips=clips)
        assert np.array_equal(to_np(d, np.uint16), exp)
        if W % 8 == 0 and (H % 8 == 0 or rot in (0, 2)):
            assert gpu.last_kernel() == "k_blend_list(fused)"
        o = to_np(out, np.uint16).reshape(N, -1)
        for f in range(N):
            assert np.array_equal(o[f], oracle.flush_frame(exp[f], W, H, rot, rot & 1)), (W, H, f)


def test_list_empty_and_errors(gpu, oracle):
    import torch
    d = torch.zeros((2, 16, 16), dtype=torch.int16, device="cuda")
    e = torch.zeros((2, 16, 16), dtype=torch.int16, device="cuda")
    ctx, other = gpu.DrawCtx(d), gpu.DrawCtx(e)
    gpu.blend_list(ctx, [])                                               # nothing to do
    out = gpu.blend_list_flush(ctx, [], 1)                                # an empty list still flushes
    assert int(out.abs().sum()) == 0
    from vvc_amd import _lib
    arr = (_lib.BlendBatch * 2)()
    arr[0] = gpu._fill_batch(ctx, gpu.BlendDsc((0, 0, 7, 7), color=0xFFFF))
    arr[1] = gpu._fill_batch(other, gpu.BlendDsc((0, 0, 7, 7), color=0xFFFF))
    rc = _lib.lib().lvgpu_blend_list_batch(arr, 2, None)
    assert rc == -1 and b"item 1" in _lib.lib().lvgpu_last_error()
    torch.cuda.synchronize()
    assert int(d.abs().sum()) == 0, "a malformed list must not draw anything"
    bad = gpu.BlendDsc((0, 0, 7, 7), src_buf=d[0, :8, :8].contiguous(), src_format=9)
    with pytest.raises(gpu.LvgpuError):
        gpu.blend_list(ctx, [gpu.BlendDsc((0, 0, 7, 7), color=0xFFFF), bad])
    torch.cuda.synchronize()
    assert int(d.abs().sum()) == 0


def test_list_full_size_deep_scene(gpu, oracle):
    """320x480 panel (480x320 logical), 256 frames, a 300-item scene: first frames against the oracle, all frames
    through a checksum of per-frame equality (every frame gets the same shared sources here)."""
    rng = np.random.default_rng(99)
    W, H, N = 480, 320, 256
    area = (0, 0, W - 1, H - 1)
    one = rng.integers(0, 65536, (1, H, W), dtype=np.uint16)
    dst = np.repeat(one, N, axis=0)
    items = [_rand_item(rng, W, H, N, True) for _ in range(300)]
    exp = _apply_oracle(oracle, one, area, items, 1, W)
    d = to_dev(dst)
    ctx = gpu.DrawCtx(d, area, area, color_16_swap=1)
    dscs, clips, keep = _gpu_items(gpu, items)
    out = gpu.blend_list_flush(ctx, dscs, 1, clips=clips)
    got = to_np(d, np.uint16)
    assert np.array_equal(got[0], exp[0])
    assert (got == got[0:1]).all()
    o = to_np(out, np.uint16).reshape(N, -1)
    assert np.array_equal(o[0], oracle.flush_frame(exp[0], W, H, 1, 0))
    assert (o == o[0:1]).all()


import glob  # noqa: E402
import os  # noqa: E402

from util import load_scene  # noqa: E402

_GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_SCENES = sorted(glob.glob(os.path.join(_GOLD, "scene*_swap*.npz")))


@pytest.mark.parametrize("path", _SCENES, ids=[os.path.basename(p)[:-4] for p in _SCENES])
def test_recorded_lvgl_scene_replayed_as_one_list(gpu, oracle, path):
    """Golden display lists recorded from the real LVGL (762-1649 descriptors per scene, mostly one-row masked fills):
    replayed over a batch of frames as ONE launch, every frame must be LVGL's own frame buffer, and the fused flush
    must be the restated flush callback's image of it."""
    swap = int(path[-5])
    fb, items = load_scene(path)
    H, W = fb.shape
    N = 6
    d = to_dev(np.random.default_rng(8).integers(0, 65536, (N, H, W), dtype=np.uint16))
    ctx = gpu.DrawCtx(d, color_16_swap=swap)
    dscs, keep = [], []
    for it in items:
        s = None if it["src"] is None else to_dev(np.ascontiguousarray(it["src"]))
        m = None if it["mask"] is None else to_dev(np.ascontiguousarray(it["mask"]))
        keep += [s, m]
        dscs.append(gpu.BlendDsc(it["area"], src_buf=s, color=it["color"], mask_buf=m, opa=it["opa"],
                                 blend_mode=it["blend_mode"]))
    out = gpu.blend_list_flush(ctx, dscs, 1, swap_bytes=0)
    assert gpu.last_kernel() == "k_blend_list(fused)"
    got = to_np(d, np.uint16)
    for f in range(N):
        assert np.array_equal(got[f], fb), "frame %d: %d pixels differ" % (f, int((got[f] != fb).sum()))
    exp_out = oracle.flush_frame(fb, W, H, 1, 0)
    o = to_np(out, np.uint16).reshape(N, -1)
    for f in range(N):
        assert np.array_equal(o[f], exp_out)


@pytest.mark.parametrize("swap", [1, 0])
@pytest.mark.parametrize("W,H,N", [(128, 96, 3), (480, 320, 2), (80, 50, 2), (72, 40, 2)])
def test_list_argb8565_destination(gpu, oracle, swap, W, H, N):
    """Display lists into the 3-byte draw buffer of a transparent screen (fill_argb / map_argb semantics, incl. the
    mask-row quirk and the blend modes); W = 72 is not a multiple of 16 and takes the per-descriptor path."""
    rng = np.random.default_rng(600 + swap + W)
    area = (0, 0, W - 1, H - 1)
    for scene in range(3):
        dst = rng.integers(0, 256, (N, H, W, 3), dtype=np.uint8)
        dst[..., 2] = rand_mask(rng, (N, H, W))
        items = []
        while len(items) < 50:
            it = _rand_item(rng, W, H, N, False)
            if it["fmt"] in (0, 1):
                items.append(it)
        if scene == 0:  # an opaque background first: every tile is written without being read
            items[0] = dict(ba=area, clip=area, opa=255, color=0x4321, src=None, mask=None, mode=0, fmt=0, mask_res=2)
        exp = dst.copy()
        for it in items:
            for f in range(N):
                oracle.blend(exp[f], area, it["clip"], it["ba"], src=None if it["src"] is None else it["src"][f],
                             color=it["color"], mask=None if it["mask"] is None else it["mask"][f], mask_res=it["mask_res"],
                             opa=it["opa"], blend_mode=it["mode"], swap=swap, screen_transp=1)
        d = to_dev(dst)
        ctx = gpu.DrawCtx(d, area, area, color_16_swap=swap, screen_transp=1)
        dscs, clips, keep = _gpu_items(gpu, items)
        gpu.blend_list(ctx, dscs, clips)
        got = to_np(d, np.uint8)
        assert np.array_equal(got, exp), "scene %d: %d bytes differ" % (scene, int((got != exp).sum()))
        if W % 16 == 0:
            assert gpu.last_kernel() == "k_blend_list"
        else:
            assert gpu.last_kernel() in ("k_blend_argb_rows", "k_blend_argb_linear")


def test_compiled_list_replay(gpu, oracle):
    """lvgpu_list_create / _replay / _destroy: the golden LVGL scene planned once, replayed (one launch, nothing
    allocated or copied) over the recorded buffers, over other buffers and over a smaller batch, with the fused flush;
    a list the tile pass cannot take whole is refused."""
    import torch
    fb, items = load_scene(os.path.join(_GOLD, "scene2_f0_swap1.npz"))
    H, W = fb.shape
    N = 5
    rng = np.random.default_rng(31)
    d = to_dev(rng.integers(0, 65536, (N, H, W), dtype=np.uint16))
    ctx = gpu.DrawCtx(d, color_16_swap=1)
    dscs, keep = [], []
    for it in items:
        s = None if it["src"] is None else to_dev(np.ascontiguousarray(it["src"]))
        m = None if it["mask"] is None else to_dev(np.ascontiguousarray(it["mask"]))
        keep += [s, m]
        dscs.append(gpu.BlendDsc(it["area"], src_buf=s, color=it["color"], mask_buf=m, opa=it["opa"],
                                 blend_mode=it["blend_mode"]))
    bl = gpu.BlendList(ctx, dscs).compile(flush=True, rotation=3)
    exp_out = oracle.flush_frame(fb, W, H, 3, 0)
    for rep in range(2):
        out = bl.replay()
        assert gpu.last_kernel() == "k_blend_list(fused)"
        got, o = to_np(d, np.uint16), to_np(out, np.uint16).reshape(N, -1)
        assert all(np.array_equal(got[f], fb) for f in range(N)) and all(np.array_equal(o[f], exp_out) for f in range(N))
        d.copy_(to_dev(rng.integers(0, 65536, (N, H, W), dtype=np.uint16)))  # scribble over the buffers between replays
    other = to_dev(rng.integers(0, 65536, (3, H, W), dtype=np.uint16))         # other buffers, smaller batch
    before = to_np(d, np.uint16).copy()
    out = bl.replay(buf=other)
    assert all(np.array_equal(to_np(other, np.uint16)[f], fb) for f in range(3))
    assert np.array_equal(to_np(out, np.uint16).reshape(3, -1)[2], exp_out)
    assert np.array_equal(to_np(d, np.uint16), before), "the recorded buffers must stay untouched"
    bl.release()
    # without flush, a translucent element on top
    bl2 = gpu.BlendList(ctx, dscs[:200]).compile()
    bl2.replay()
    assert gpu.last_kernel() == "k_blend_list"
    bl2.release()
    # a translucent unmasked fill whose value on black depends on scan order cannot be compiled
    bad = gpu.BlendList(ctx, [gpu.BlendDsc((0, 0, W - 1, H - 1), color=1), gpu.BlendDsc((3, 3, 40, 40), color=0x00F0, opa=135)])
    with pytest.raises(gpu.LvgpuError) as ei:
        bad.compile()
    assert ei.value.code == -3
    # nor can a buffer whose width is off the 8-pixel grid
    d7 = torch.zeros((2, 16, 20), dtype=torch.int16, device="cuda")
    with pytest.raises(gpu.LvgpuError):
        gpu.BlendList(gpu.DrawCtx(d7), [gpu.BlendDsc((0, 0, 19, 15), color=1)]).compile()


def test_compiled_list_replay_frame_count(gpu, oracle):
    """A compiled list whose items read per-frame sources or masks refuses a replay over more frames than were recorded
    (frame n - 1 would read past those buffers); a list of shared inputs takes any number of frames."""
    rng = np.random.default_rng(33)
    W, H, N = 64, 64, 3
    area = (0, 0, W - 1, H - 1)
    d = to_dev(rng.integers(0, 65536, (N, H, W), dtype=np.uint16))
    ctx = gpu.DrawCtx(d, color_16_swap=1)
    per_frame = to_dev(rand_mask(rng, (N, H, W)))
    shared = to_dev(rand_mask(rng, (H, W)))
    big = to_dev(rng.integers(0, 65536, (N + 4, H, W), dtype=np.uint16))
    bl = gpu.BlendList(ctx, [gpu.BlendDsc(area, color=0x1234, mask_buf=per_frame, opa=255)]).compile()
    bl.replay()                                      # the recorded batch
    bl.replay(buf=big[:2])                           # fewer frames
    with pytest.raises(gpu.LvgpuError) as ei:
        bl.replay(buf=big)                           # more frames than the per-frame mask holds
    assert ei.value.code == -1 and "recorded for 3 frames" in str(ei.value)
    bl.release()
    bl = gpu.BlendList(ctx, [gpu.BlendDsc(area, color=0x1234, mask_buf=shared, opa=255)]).compile()
    before = to_np(big, np.uint16).copy()
    bl.replay(buf=big)
    exp = before.copy()
    m = to_np(shared, np.uint8)
    for f in range(N + 4):
        oracle.blend(exp[f], area, area, area, color=0x1234, mask=m, opa=255, swap=1)
    assert np.array_equal(to_np(big, np.uint16), exp)
    bl.release()
