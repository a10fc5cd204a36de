"""CPU: fuzz the oracle against the REAL LVGL (oracle/_ref, compiled from the reference sources).  Skipped where the
reference build is not present; the committed golden vectors cover that case."""
import numpy as np
import pytest

from util import OPAS, rand_mask


@pytest.fixture(scope="module", params=[1, 0])
def ref(request, oracle):
    L = oracle.ref(request.param)
    if L is None:
        pytest.skip("oracle/_ref not built")
    return request.param, L


def test_fuzz_blend_normal(oracle, ref):
    swap, L = ref
    rng = np.random.default_rng(swap)
    W, H = 67, 9
    for t in range(4000):
        buf = rng.integers(0, 65536, (H, W), dtype=np.uint16)
        if t % 3 == 0:
            buf[rng.random((H, W)) < 0.6] = 0
        x1, y1 = int(rng.integers(-3, W - 2)), int(rng.integers(-2, H - 1))
        x2, y2 = int(rng.integers(x1, W + 3)), int(rng.integers(y1, H + 2))
        cx1, cy1 = int(rng.integers(0, W)), int(rng.integers(0, H))
        clip = (cx1, cy1, int(rng.integers(cx1, W)), int(rng.integers(cy1, H)))
        bw, bh = x2 - x1 + 1, y2 - y1 + 1
        src = rng.integers(0, 65536, (bh, bw), dtype=np.uint16) if rng.random() < 0.5 else None
        ma = (x1 - int(rng.integers(0, 3)), y1 - int(rng.integers(0, 3)), x2 + int(rng.integers(0, 3)),
              y2 + int(rng.integers(0, 3)))
        mask = rand_mask(rng, (ma[3] - ma[1] + 1, ma[2] - ma[0] + 1)) if rng.random() < 0.6 else None
        opa = int(rng.choice(OPAS))
        mres = int(rng.choice([0, 1, 2, 2, 2, 3]))
        color = int(rng.integers(0, 65536))
        mode = int(rng.choice([0, 0, 0, 1, 2, 3, 4]))
        b1, b2 = buf.copy(), buf.copy()
        oracle.ref_blend(L, b1, (0, 0, W - 1, H - 1), clip, (x1, y1, x2, y2), src=src, color=color, mask=mask,
                         mask_res=mres, mask_area=ma, opa=opa, blend_mode=mode)
        oracle.blend(b2, (0, 0, W - 1, H - 1), clip, (x1, y1, x2, y2), src=src, color=color, mask=mask, mask_res=mres,
                     mask_area=ma, opa=opa, blend_mode=mode, swap=swap)
        assert np.array_equal(b1, b2), (t, opa, mres, mode)


def test_fuzz_blend_argb8565_destination(oracle, ref):
    swap, L = ref
    rng = np.random.default_rng(10 + swap)
    W, H = 41, 7
    for t in range(2500):
        buf = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        buf[..., 2] = rand_mask(rng, (H, W))
        x1, y1 = int(rng.integers(-2, W - 2)), int(rng.integers(-2, H - 1))
        x2, y2 = int(rng.integers(x1, W + 2)), int(rng.integers(y1, H + 2))
        cx1, cy1 = int(rng.integers(0, W)), int(rng.integers(0, H))
        clip = (cx1, cy1, int(rng.integers(cx1, W)), int(rng.integers(cy1, H)))
        bw, bh = x2 - x1 + 1, y2 - y1 + 1
        src = rng.integers(0, 65536, (bh, bw), dtype=np.uint16) if rng.random() < 0.5 else None
        mask = rand_mask(rng, (bh, bw)) if rng.random() < 0.6 else None
        opa = int(rng.choice(OPAS))
        # map_argb keeps function-static memo state keyed without the blend function, so only one non-normal
        # mode is exercised per library instance (any other order would read another mode's memo)
        mode = int(rng.choice([0, 0, 1])) if src is not None else 0
        color = int(rng.integers(0, 65536))
        b1, b2 = buf.copy(), buf.copy()
        oracle.ref_blend(L, b1, (0, 0, W - 1, H - 1), clip, (x1, y1, x2, y2), src=src, color=color, mask=mask,
                         opa=opa, blend_mode=mode, screen_transp=1)
        oracle.blend(b2, (0, 0, W - 1, H - 1), clip, (x1, y1, x2, y2), src=src, color=color, mask=mask, opa=opa,
                     blend_mode=mode, swap=swap, screen_transp=1)
        assert np.array_equal(b1, b2), (t, opa, mode, src is not None, mask is not None)


def test_color_functions(oracle, ref):
    swap, L = ref
    rng = np.random.default_rng(3)
    n = 300000
    c1 = rng.integers(0, 65536, n, dtype=np.uint16)
    c2 = rng.integers(0, 65536, n, dtype=np.uint16)
    mix = rng.integers(0, 256, n, dtype=np.uint8)
    a = np.empty(n, np.uint16)
    b = np.empty(n, np.uint16)
    L.ref_color_mix_n(oracle._p(c1), oracle._p(c2), oracle._p(mix), oracle._p(a), n)
    oracle.lib().lvref_color_mix_n(oracle._p(c1), oracle._p(c2), oracle._p(mix), oracle._p(b), n, swap)
    assert np.array_equal(a, b)
    for mx in (0, 127, 255):
        assert L.ref_color_mix_digest(mx, 1000, 1100) == oracle.lib().lvref_color_mix_digest(mx, 1000, 1100, swap)


def test_image_with_alpha_draw(oracle, ref):
    swap, L = ref
    rng = np.random.default_rng(4)
    for (W, H, clip, coords, opa) in [(160, 50, (0, 0, 159, 49), (0, 0, 159, 49), 255),
                                      (160, 50, (5, 6, 100, 40), (2, 3, 120, 45), 200),
                                      (33, 17, (0, 0, 32, 16), (0, 0, 32, 16), 253)]:
        iw, ih = coords[2] - coords[0] + 1, coords[3] - coords[1] + 1
        src = rng.integers(0, 256, (ih, iw, 3), dtype=np.uint8)
        src[..., 2] = rand_mask(rng, (ih, iw))
        dst = rng.integers(0, 65536, (H, W), dtype=np.uint16)
        d1, d2 = dst.copy(), dst.copy()
        L.ref_init(W, H)
        area = np.array([0, 0, W - 1, H - 1], np.int16)
        L.ref_img_decoded(oracle._p(d1), oracle._p(area), oracle._p(np.array(clip, np.int16)),
                          oracle._p(np.array(coords, np.int16)), oracle._p(src), 5, opa, 0, 0, 0)
        L.ref_init(480, 480)
        oracle.img_blend(d2, (0, 0, W - 1, H - 1), clip, coords, src, oracle.SRC_ARGB8565, opa, swap, W)
        assert np.array_equal(d1, d2)


def _lvgl_image_draw(oracle, L, dst, W, H, coords, src, cf, opa, recolor, recolor_opa):
    L.ref_init(W, H)
    area = np.array([0, 0, W - 1, H - 1], np.int16)
    # lv_draw_img hands lv_draw_sw_img_decoded a clip area that lies within the image (lv_draw_img.c:236-262)
    c16 = np.array(coords, np.int16)
    L.ref_img_decoded(oracle._p(dst), oracle._p(area), oracle._p(c16), oracle._p(c16), oracle._p(src), cf, opa, recolor,
                      recolor_opa, 0)
    L.ref_init(480, 480)


def test_image_convert_recolor_chroma_key(oracle, ref):
    """convert_cb + the recolor pass of lv_draw_sw_img_decoded (lv_draw_sw_img.c:163-174, :203-298): plain, chroma-keyed
    and alpha images, recolored or not, drawn by the real LVGL; the oracle converts the image (colour + alpha plane)
    and blends the planes."""
    swap, L = ref
    rng = np.random.default_rng(21)
    key = int(oracle.lib().lvref_color_hex(0x00FF00, swap))  # LV_COLOR_CHROMA_KEY, src/lv_conf.h:42
    W, H = 96, 40
    coords = (3, 2, 90, 35)
    iw, ih = coords[2] - coords[0] + 1, coords[3] - coords[1] + 1
    area = (0, 0, W - 1, H - 1)
    for cf, fmt in ((4, 1), (6, 1), (5, 4)):  # LV_IMG_CF_TRUE_COLOR, _CHROMA_KEYED, _ALPHA
        for recolor_opa in (0, 2, 3, 128, 250, 255):
            for opa in (255, 180):
                if fmt == 1:
                    src = rng.integers(0, 65536, (ih, iw), dtype=np.uint16)
                    src[rng.random((ih, iw)) < 0.3] = key
                else:
                    src = rng.integers(0, 256, (ih, iw, 3), dtype=np.uint8)
                    src[..., 2] = rand_mask(rng, (ih, iw))
                recolor = int(rng.integers(0, 65536))
                dst = rng.integers(0, 65536, (H, W), dtype=np.uint16)
                d1, d2 = dst.copy(), dst.copy()
                _lvgl_image_draw(oracle, L, d1, W, H, coords, src, cf, opa, recolor, recolor_opa)
                color, alpha = oracle.image_convert(src, fmt, swap, chroma_keyed=int(cf == 6), chroma_key=key,
                                                    recolor=recolor, recolor_opa=recolor_opa)
                # a plain image without recolor is blended without any mask (:65-70); otherwise the alpha plane is the
                # mask, with mask_res FULL_COVER (= ignored) for LV_IMG_CF_TRUE_COLOR (:128-131)
                mask = None if cf == 4 else alpha.reshape(ih, iw)
                oracle.blend(d2, area, area, coords, src=color.reshape(ih, iw), mask=mask, opa=opa, swap=swap)
                assert np.array_equal(d1, d2), (cf, recolor_opa, opa)


def test_fuzz_blend_without_antialiasing(oracle, ref):
    """disp->driver->antialiasing == 0: lv_draw_sw_blend_basic rounds the caller's mask in place over the whole
    mask_area (lv_draw_sw_blend.c:150-156) -- draw buffer AND mask must come out as LVGL leaves them, for RGB565 and
    ARGB8565 destinations; a TRUE_COLOR_ALPHA image has its alpha plane rounded chunk by chunk."""
    swap, L = ref
    rng = np.random.default_rng(70 + swap)
    W, H = 67, 9
    full = (0, 0, W - 1, H - 1)
    L.ref_set_antialiasing(0)
    try:
        for t in range(3000):
            transp = int(t % 3 == 0)
            buf = rng.integers(0, 256, (H, W, 3), dtype=np.uint8) if transp else rng.integers(0, 65536, (H, W), dtype=np.uint16)
            x1, y1 = int(rng.integers(-3, W - 2)), int(rng.integers(-2, H - 1))
            x2, y2 = int(rng.integers(x1, W + 3)), int(rng.integers(y1, H + 2))
            cx1, cy1 = int(rng.integers(0, W)), int(rng.integers(0, H))
            clip = (cx1, cy1, int(rng.integers(cx1, W)), int(rng.integers(cy1, H)))
            bw, bh = x2 - x1 + 1, y2 - y1 + 1
            src = rng.integers(0, 65536, (bh, bw), dtype=np.uint16) if rng.random() < 0.5 else None
            ma = (x1 - int(rng.integers(0, 3)), y1 - int(rng.integers(0, 3)), x2 + int(rng.integers(0, 3)),
                  y2 + int(rng.integers(0, 3)))
            mask = rng.integers(120, 137, (ma[3] - ma[1] + 1, ma[2] - ma[0] + 1), dtype=np.uint8) if t % 2 else \
                rand_mask(rng, (ma[3] - ma[1] + 1, ma[2] - ma[0] + 1))
            opa = int(rng.choice(OPAS))
            mres = int(rng.choice([1, 2, 2, 2, 3]))
            color = int(rng.integers(0, 65536))
            mode = int(rng.choice([0, 0, 0, 1, 2, 3]))
            b1, b2 = buf.copy(), buf.copy()
            m1, m2 = mask.copy(), mask.copy()
            L.ref_blend(oracle._p(b1), oracle._p(oracle._area16(full)), oracle._p(oracle._area16(clip)),
                        oracle._p(oracle._area16((x1, y1, x2, y2))), oracle._p(src), color, oracle._p(m1), mres,
                        oracle._p(oracle._area16(ma)), opa, mode, transp)
            oracle.blend(b2, full, clip, (x1, y1, x2, y2), src=src, color=color, mask=m2, mask_res=mres, mask_area=ma, opa=opa,
                         blend_mode=mode, swap=swap, screen_transp=transp, antialiasing=0)
            assert np.array_equal(b1, b2), (t, opa, mres, mode, transp)
            assert np.array_equal(m1, m2), (t, "mask as the call leaves it")
        for t in range(200):
            buf = rng.integers(0, 65536, (H, W), dtype=np.uint16)
            co = (int(rng.integers(0, 10)), int(rng.integers(0, 3)), int(rng.integers(30, W)), int(rng.integers(4, H)))
            clip = (co[0] + int(rng.integers(0, 3)), co[1], co[2] - int(rng.integers(0, 3)), co[3])
            src = rng.integers(0, 256, (co[3] - co[1] + 1, co[2] - co[0] + 1, 3), dtype=np.uint8)
            opa = int(rng.choice([255, 254, 253, 128]))
            b1, b2 = buf.copy(), buf.copy()
            L.ref_img_decoded(oracle._p(b1), oracle._p(oracle._area16(full)), oracle._p(oracle._area16(clip)),
                              oracle._p(oracle._area16(co)), oracle._p(src), 5, opa, 0, 0, 0)
            oracle.img_blend(b2, full, clip, co, src, 4, opa, swap, 480, antialiasing=0)
            assert np.array_equal(b1, b2), (t, opa)
    finally:
        L.ref_set_antialiasing(1)
