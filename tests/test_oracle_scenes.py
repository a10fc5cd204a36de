"""CPU: the oracle replays the golden display lists -- every blend descriptor the real LVGL's producers issued for a
widget scene (rectangles with radius/gradient/border/shadow, labels, arc, bar, button, line, alpha images) -- and
must arrive at the frame buffer LVGL itself drew.  Pins the restated blend against LVGL on real producer output
(mostly one-row masked fills) without needing oracle/_ref at test time."""
import glob
import os

import numpy as np
import pytest

from util import load_scene

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SCENES = sorted(glob.glob(os.path.join(GOLD, "scene*_swap*.npz")))


def test_fixtures_present():
    assert len(SCENES) == 8


@pytest.mark.parametrize("path", SCENES, ids=[os.path.basename(p)[:-4] for p in SCENES])
def test_oracle_replays_recorded_scene(oracle, path):
    swap = int(path[-5])
    fb, items = load_scene(path)
    H, W = fb.shape
    assert len(items) > 500
    buf = np.random.default_rng(3).integers(0, 65536, (H, W), dtype=np.uint16)  # the scene repaints every pixel
    area = (0, 0, W - 1, H - 1)
    for it in items:
        oracle.blend(buf, area, it["area"], it["area"], src=it["src"], color=it["color"], mask=it["mask"], opa=it["opa"],
                     blend_mode=it["blend_mode"], swap=swap)
    assert np.array_equal(buf, fb)


def test_live_recording_matches_fixture(oracle):
    """Where oracle/_ref is present, a fresh recording must equal the committed fixture (the generator is reproducible)."""
    L = oracle.ref(1)
    if L is None or not hasattr(L, "ref_record_items"):
        pytest.skip("oracle/_ref (real LVGL with the recorder) not present")
    fb, items = oracle.recorded_scene(L, 480, 320, 2, 0)
    gfb, gitems = load_scene(os.path.join(GOLD, "scene2_f0_swap1.npz"))
    assert np.array_equal(fb, gfb) and len(items) == len(gitems)
    for a, b in zip(items, gitems):
        assert tuple(a["area"]) == tuple(b["area"]) and a["opa"] == b["opa"] and a["color"] == b["color"]
        assert (a["mask"] is None) == (b["mask"] is None) and (a["mask"] is None or np.array_equal(a["mask"], b["mask"]))
