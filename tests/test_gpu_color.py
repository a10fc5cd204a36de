"""GPU parity for the colour-conversion, split, fill, copy, synthetic-data and checksum entry points."""
import numpy as np
import pytest
import torch

from util import to_dev, to_np

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("swap", [1, 0])
def test_color_from_rgb888_all_inputs(gpu, oracle, swap):
    allhex = np.arange(1 << 24, dtype=np.uint32)
    got = to_np(gpu.color_from_rgb888(to_dev(allhex), swap), np.uint16)
    exp = np.empty(1 << 24, np.uint16)
    oracle.lib().lvref_color_hex_n(oracle._p(allhex), oracle._p(exp), 1 << 24, swap)
    assert np.array_equal(got, exp)


@pytest.mark.parametrize("swap", [1, 0])
def test_color_to32_all_inputs(gpu, oracle, swap):
    all16 = np.arange(1 << 16, dtype=np.uint16)
    got = to_np(gpu.color_to32(to_dev(all16), swap), np.uint32)
    exp = np.empty(1 << 16, np.uint32)
    oracle.lib().lvref_color_to32_n(oracle._p(all16), oracle._p(exp), 1 << 16, swap)
    assert np.array_equal(got, exp)


@pytest.mark.parametrize("swap", [1, 0])
@pytest.mark.parametrize("fmt", [2, 3, 4])
def test_image_split(gpu, oracle, swap, fmt):
    rng = np.random.default_rng(fmt)
    n = 100003
    src = rng.integers(0, 256, n * 3, dtype=np.uint8) if fmt == 4 else rng.integers(0, 1 << 32, n, dtype=np.uint32)
    c, a = gpu.image_split(to_dev(src), fmt, swap)
    ec, ea = oracle.image_split(src, fmt, swap)
    assert np.array_equal(to_np(c, np.uint16), ec) and np.array_equal(a.cpu().numpy(), ea)


def test_color_fill_and_buffer_copy(gpu):
    buf = torch.zeros(12345, dtype=torch.int16, device="cuda")
    gpu.color_fill(buf, 0xA5C3)
    assert bool((buf.view(torch.int16) == np.int16(np.uint16(0xA5C3).view(np.int16))).all())
    # lv_color_fill over every start alignment and length around the 16-byte words: nothing outside [lo, lo + n) changes
    guard = np.int16(0x1357)
    c = np.int16(np.uint16(0xBEEF).view(np.int16))
    for lo in range(0, 9):
        for n in (1, 2, 7, 8, 9, 15, 16, 17, 63, 64, 65, 1000, 4099):
            big = torch.full((lo + n + 40,), int(guard), dtype=torch.int16, device="cuda")
            gpu.color_fill(big[lo:lo + n], 0xBEEF)
            h = big.cpu().numpy()
            assert (h[:lo] == guard).all() and (h[lo + n:] == guard).all() and (h[lo:lo + n] == c).all(), (lo, n)
    rng = np.random.default_rng(1)
    N = 3
    src = rng.integers(0, 65536, (N, 20, 33), dtype=np.uint16)
    dst = rng.integers(0, 65536, (N, 40, 50), dtype=np.uint16)
    d = to_dev(dst)
    # lv_draw_sw_buffer_copy: width and row count come from dest_area, the source origin from src_area
    gpu.buffer_copy(d, (5, 7, 24, 18), to_dev(src), (3, 2, 22, 13))
    exp = dst.copy()
    exp[:, 7:19, 5:25] = src[:, 2:14, 3:23]
    assert np.array_equal(to_np(d, np.uint16), exp)


def test_synth_and_checksum_match_cpu_definition(gpu, oracle):
    for kind, dt in ((gpu.SYNTH_U16, np.uint16), (gpu.SYNTH_MASK_UI, np.uint8), (gpu.SYNTH_MASK_UNI, np.uint8),
                     (gpu.SYNTH_ARGB_UI, np.uint32), (gpu.SYNTH_ARGB_UNI, np.uint32)):
        g = gpu.synth(kind, 0x5EED0001, 12345678901, (70001,))
        h = oracle.synth(kind, 0x5EED0001, 12345678901, 70001)
        assert np.array_equal(to_np(g, dt), h)
        assert gpu.checksum(g, 777) == oracle.checksum(h, 777)
    m = oracle.synth(oracle.SYNTH_MASK_UI, 3, 0, 300000)
    assert abs((m == 0).mean() - 1 / 3) < 0.01 and abs((m == 255).mean() - 1 / 3) < 0.01
    # shards of one stream concatenate to the stream (what makes any frame partition reproducible)
    a = gpu.synth(gpu.SYNTH_U16, 5, 0, (1000,))
    b = gpu.synth(gpu.SYNTH_U16, 5, 1000, (500,))
    ab = gpu.synth(gpu.SYNTH_U16, 5, 0, (1500,))
    assert torch.equal(torch.cat([a, b]), ab)
    assert (gpu.checksum(a, 0) + gpu.checksum(b, 1000)) & 0xFFFFFFFFFFFFFFFF == gpu.checksum(ab, 0)


@pytest.mark.parametrize("swap", [1, 0])
def test_image_convert_recolor_chroma_key(gpu, oracle, swap):
    """lvgpu_image_convert against the oracle's convert_cb + recolor on random images, and end to end against the
    buffers the real LVGL drew (golden): GPU-converted planes blended on the GPU."""
    import os
    import torch
    rng = np.random.default_rng(5 + swap)
    key = int(oracle.lib().lvref_color_hex(0x00FF00, swap))
    for fmt in (1, 2, 3, 4):
        for recolor_opa in (0, 2, 3, 77, 255):
            n = 5000
            if fmt == 1:
                src = rng.integers(0, 65536, n, dtype=np.uint16)
                src[rng.random(n) < 0.2] = key
            elif fmt == 4:
                src = rng.integers(0, 256, (n, 3), dtype=np.uint8)
            else:
                src = rng.integers(0, 1 << 32, n, dtype=np.uint32)
            recolor = int(rng.integers(0, 65536))
            for keyed in ((0, 1) if fmt == 1 else (0,)):
                ec, ea = oracle.image_convert(src, fmt, swap, chroma_keyed=keyed, chroma_key=key, recolor=recolor,
                                              recolor_opa=recolor_opa)
                gc, ga = gpu.image_convert(to_dev(src), fmt, swap, chroma_key=key if keyed else None, recolor=recolor,
                                           recolor_opa=recolor_opa)
                assert np.array_equal(to_np(gc, np.uint16), ec) and np.array_equal(to_np(ga, np.uint8), ea), (fmt, recolor_opa)
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lvgl_image_recolor_swap%d.npz" % swap))
    coords = tuple(int(v) for v in z["coords"])
    gkey = int(z["chroma_key"][0])
    iw, ih = coords[2] - coords[0] + 1, coords[3] - coords[1] + 1
    for k, (cf, recolor, recolor_opa, opa) in enumerate(z["cases"].tolist()):
        src, dst, out = z["src%d" % k], z["dst%d" % k], z["out%d" % k]
        c, a = gpu.image_convert(to_dev(src), 4 if cf == 5 else 1, swap, chroma_key=gkey if cf == 6 else None,
                                 recolor=recolor, recolor_opa=recolor_opa)
        d = to_dev(np.repeat(dst[None], 2, axis=0))
        ctx = gpu.DrawCtx(d, clip_area=coords, color_16_swap=swap)
        dsc = gpu.BlendDsc(coords, src_buf=c.view(ih, iw), mask_buf=None if cf == 4 else a.view(ih, iw), opa=opa)
        gpu.lv_draw_sw_blend(ctx, dsc)
        got = to_np(d, np.uint16)
        assert np.array_equal(got[0], out) and np.array_equal(got[1], out), (k, cf, recolor_opa, opa)


@pytest.mark.parametrize("swap", [1, 0])
def test_conversions_at_any_alignment_and_length(gpu, oracle, swap):
    """The conversion entry points take 8 pixels per lane when their buffers are 16-byte aligned, pixel by pixel
    otherwise and for what is left over: every combination of misaligned input / output and of lengths around a
    multiple of 8 gives the oracle's values and leaves the bytes beside the output alone."""
    from vvc_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(17 + swap)
    for n in (1, 7, 8, 9, 64, 1001, 4099):
        for in_off, out_off in ((0, 0), (1, 0), (0, 1), (3, 5)):
            x = rng.integers(0, 1 << 24, n + in_off, dtype=np.uint32)
            exp = np.empty(n, np.uint16)
            oracle.lib().lvref_color_hex_n(oracle._p(x[in_off:].copy()), oracle._p(exp), n, swap)
            dx = to_dev(x)
            out = torch.full((n + out_off + 9,), 0x1111, dtype=torch.int16, device="cuda")
            assert L.lvgpu_color_from_rgb888(dx.data_ptr() + 4 * in_off, out.data_ptr() + 2 * out_off, n, swap, None) == 0
            h = to_np(out, np.uint16)
            assert np.array_equal(h[out_off:out_off + n], exp) and (h[:out_off] == 0x1111).all() and (h[out_off + n:] == 0x1111).all()
            c = rng.integers(0, 65536, n + in_off, dtype=np.uint16)
            exp32 = np.empty(n, np.uint32)
            oracle.lib().lvref_color_to32_n(oracle._p(c[in_off:].copy()), oracle._p(exp32), n, swap)
            out32 = torch.full((n + out_off + 9,), 0x22222222, dtype=torch.int32, device="cuda")
            assert L.lvgpu_color_to32(to_dev(c).data_ptr() + 2 * in_off, out32.data_ptr() + 4 * out_off, n, swap, None) == 0
            h32 = to_np(out32, np.uint32)
            assert np.array_equal(h32[out_off:out_off + n], exp32) and (h32[:out_off] == 0x22222222).all() and \
                (h32[out_off + n:] == 0x22222222).all()
            for fmt in (1, 2, 4):
                if fmt == 1:
                    src = rng.integers(0, 65536, n + in_off, dtype=np.uint16)
                    sview, sbytes = src[in_off:], 2 * in_off
                elif fmt == 2:
                    src = rng.integers(0, 1 << 32, n + in_off, dtype=np.uint32)
                    sview, sbytes = src[in_off:], 4 * in_off
                else:
                    src = rng.integers(0, 256, (n + in_off, 3), dtype=np.uint8)
                    sview, sbytes = src[in_off:], 3 * in_off
                ec, ea = oracle.image_convert(np.ascontiguousarray(sview), fmt, swap, recolor=0x3D2A, recolor_opa=90)
                col = torch.full((n + out_off + 9,), 0x1111, dtype=torch.int16, device="cuda")
                alp = torch.full((n + out_off + 9,), 0x33, dtype=torch.uint8, device="cuda")
                o = _lib.ImageOpts()
                o.recolor, o.recolor_opa = 0x3D2A, 90
                import ctypes as C
                assert L.lvgpu_image_convert(to_dev(src).data_ptr() + sbytes, fmt, col.data_ptr() + 2 * out_off,
                                             alp.data_ptr() + out_off, n, swap, C.byref(o), None) == 0
                hc, ha = to_np(col, np.uint16), to_np(alp, np.uint8)
                assert np.array_equal(hc[out_off:out_off + n], ec) and np.array_equal(ha[out_off:out_off + n], ea), (n, in_off, out_off, fmt)
                assert (hc[:out_off] == 0x1111).all() and (hc[out_off + n:] == 0x1111).all()
                assert (ha[:out_off] == 0x33).all() and (ha[out_off + n:] == 0x33).all()
