"""CPU: the restated flush callback (src/lv_port.c:428-572).  The reference file cannot be compiled in the build
image (it needs ESP-IDF / FreeRTOS headers), so this row has no golden vectors from the reference itself: "parity
unpinned".  What is checked: the chunked loop nest against the closed-form mapping, and the chunk rectangles
against the values recorded when the unmodified lv_port.c was executed during the survey of the reference
(ROT_90, full 480x320 frame, trans_size = 320*480/10: ten draw_bitmap calls x in [0,320), y in [48k, 48k+48))."""
import numpy as np
import pytest


@pytest.mark.parametrize("rot", [0, 1, 2, 3])
@pytest.mark.parametrize("shape", [(480, 320), (320, 480), (100, 37), (33, 7), (8, 8), (1, 5)])
def test_chunks_concatenate_to_closed_form(oracle, rot, shape):
    W, H = shape
    rng = np.random.default_rng(W + 31 * H + rot)
    frame = rng.integers(0, 65536, W * H, dtype=np.uint16)
    trans = max(W * H // 10, max(W, H))
    chunks = oracle.flush((0, 0, W - 1, H - 1), frame, rot, trans, W, H)
    cat = np.concatenate([c[4] for c in chunks])
    assert np.array_equal(cat, oracle.flush_frame(frame, W, H, rot))
    # the chunks tile the panel exactly once
    pw, ph = (H, W) if rot in (1, 3) else (W, H)
    cover = np.zeros((ph, pw), np.int32)
    panel = np.zeros((ph, pw), np.uint16)
    for x0, y0, x1, y1, data in chunks:
        cover[y0:y1, x0:x1] += 1
        panel[y0:y1, x0:x1] = data.reshape(y1 - y0, x1 - x0)
    assert (cover == 1).all()
    assert np.array_equal(panel.reshape(-1), oracle.flush_frame(frame, W, H, rot))


def test_rot90_chunk_rectangles_of_the_firmware_geometry(oracle):
    W, H = 480, 320  # logical LVGL frame under ROT_90 (src/esp_bsp.c:343-350); panel is 320x480
    frame = np.arange(W * H, dtype=np.uint32).astype(np.uint16)
    chunks = oracle.flush((0, 0, W - 1, H - 1), frame, 1, 320 * 480 // 10, W, H)
    assert len(chunks) == 10
    for k, (x0, y0, x1, y1, data) in enumerate(chunks):
        assert (x0, x1) == (0, 320) and (y0, y1) == (48 * k, 48 * k + 48)
        assert data.size == 48 * 320
    # the shipped firmware runs ROT_NONE on a 320x480 frame: ten 48-row chunks
    chunks = oracle.flush((0, 0, 319, 479), frame, 0, 320 * 480 // 10, 320, 480)
    assert [(c[0], c[1], c[2], c[3]) for c in chunks] == [(0, 48 * k, 320, 48 * k + 48) for k in range(10)]


def test_mapping_formulas(oracle):
    W, H = 12, 5
    frame = np.arange(W * H, dtype=np.uint16)
    img = frame.reshape(H, W)
    assert np.array_equal(oracle.flush_frame(frame, W, H, 0).reshape(H, W), img)
    assert np.array_equal(oracle.flush_frame(frame, W, H, 2).reshape(H, W), img[::-1, ::-1])
    assert np.array_equal(oracle.flush_frame(frame, W, H, 1).reshape(W, H), img[::-1, :].T)   # out[x][H-1-y]
    assert np.array_equal(oracle.flush_frame(frame, W, H, 3).reshape(W, H), img[:, ::-1].T)   # out[W-1-x][y]
    assert np.array_equal(oracle.flush_frame(frame, W, H, 0, 1), frame.byteswap())


@pytest.mark.parametrize("swap", [1, 0])
def test_rotation_mappings_against_lvgl_own_sw_rotate(oracle, swap):
    """The reference's flush callback cannot be compiled here, but LVGL carries its own software rotation
    (libraries/lvgl/src/core/lv_refr.c:1084-1265) for the same three permutations, with the two quarter turns named the
    other way round.  Golden frames: a widget scene rendered un-rotated and what LVGL itself flushed with
    drv.sw_rotate for drv.rotated = 90/180/270 (oracle/gen_golden_scenes.py)."""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lvgl_sw_rotate_swap%d.npz" % swap))
    for LW, LH in ((480, 320), (320, 480), (200, 96)):
        logical = z["logical_%dx%d" % (LW, LH)]
        assert logical.shape == (LH, LW) and len(np.unique(logical)) > 20
        for lv_rot, mine in ((1, 3), (2, 2), (3, 1)):
            lvgl = z["lv_rot%d_%dx%d" % (lv_rot, LW, LH)]
            got = oracle.flush_frame(logical, LW, LH, mine, 0)
            assert np.array_equal(got, lvgl.ravel()), (LW, LH, lv_rot)
            other = oracle.flush_frame(logical, LW, LH, lv_rot, 0)
            if lv_rot != 2:
                assert not np.array_equal(other, lvgl.ravel()), "the two namings must differ"
