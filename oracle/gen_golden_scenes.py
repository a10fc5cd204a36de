"""TEST INFRASTRUCTURE.  Golden display lists: the real LVGL (oracle/_ref, built from the reference's sources) renders
the widget scenes of oracle/ref_harness.c while a wrapped draw context records every blend descriptor its producers
issue; the descriptors (inputs) and LVGL's frame buffer (expected output) are stored as
tests/golden/scene<id>_f<flags>_swap<s>.npz.  Run here, where /root/reference exists:  python -m oracle.gen_golden_scenes
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import lvref_py as O  # noqa: E402

GOLD = os.path.join(os.path.dirname(HERE), "tests", "golden")
W, H = 480, 320


def pack(fb, items):
    n = len(items)
    meta = np.zeros((n, 9), np.int32)  # x1 y1 x2 y2 color opa blend_mode has_src has_mask
    srcs, masks = [], []
    for i, it in enumerate(items):
        meta[i] = list(it["area"]) + [it["color"], it["opa"], it["blend_mode"], it["src"] is not None, it["mask"] is not None]
        if it["src"] is not None:
            srcs.append(it["src"].ravel())
        if it["mask"] is not None:
            masks.append(it["mask"].ravel())
    return dict(fb=fb, meta=meta, src=np.concatenate(srcs) if srcs else np.zeros(0, np.uint16),
                mask=np.concatenate(masks) if masks else np.zeros(0, np.uint8))


def main():
    for swap in (1, 0):
        L = O.ref(swap)
        assert L is not None, "oracle/_ref is not built (make -C oracle ref)"
        for scene, flags in ((0, 0), (1, 0), (2, 0), (2, 1)):
            fb, items = O.recorded_scene(L, W, H, scene, flags)
            path = os.path.join(GOLD, "scene%d_f%d_swap%d.npz" % (scene, flags, swap))
            np.savez_compressed(path, **pack(fb, items))
            print(path, len(items), "descriptors", os.path.getsize(path), "bytes")
        # LVGL's own software rotation (drv.sw_rotate, libraries/lvgl/src/core/lv_refr.c:1084-1265) of whole frames:
        # the un-rotated logical frame and what LVGL flushes to a panel for drv.rotated = 90/180/270.
        arrays = {}
        for LW, LH in ((480, 320), (320, 480), (200, 96)):
            arrays["logical_%dx%d" % (LW, LH)] = O.render_scene(L, LW, LH, 2, 0)
            for r in (1, 2, 3):
                hor, ver = (LW, LH) if r == 2 else (LH, LW)
                arrays["lv_rot%d_%dx%d" % (r, LW, LH)] = O.render_scene(L, hor, ver, 2, 4 | (r << 3))
        L.ref_scene_display(16, 16, None, None, 0, None, None, 1, 0)
        # images drawn by LVGL with recolor / chroma key (lv_draw_sw_img.c:163-174, :203-298): inputs and LVGL's buffer
        img = {}
        rng = np.random.default_rng(2024)
        key = int(O.lib().lvref_color_hex(0x00FF00, swap))  # LV_COLOR_CHROMA_KEY of the firmware configuration
        IW, IH, coords = 96, 40, (3, 2, 90, 35)
        iw, ih = coords[2] - coords[0] + 1, coords[3] - coords[1] + 1
        c16 = np.array(coords, np.int16)
        full = np.array([0, 0, IW - 1, IH - 1], np.int16)
        cases = []
        for cf in (4, 6, 5):
            for recolor_opa, opa in ((0, 255), (3, 255), (128, 180), (255, 255)):
                if cf == 5:
                    src = rng.integers(0, 256, (ih, iw, 3), dtype=np.uint8)
                    src[..., 2] = rng.choice(np.array([0, 255, 1, 2, 127, 128, 253, 254], np.uint8), (ih, iw))
                else:
                    src = rng.integers(0, 65536, (ih, iw), dtype=np.uint16)
                    src[rng.random((ih, iw)) < 0.3] = key
                recolor = int(rng.integers(0, 65536))
                dst = rng.integers(0, 65536, (IH, IW), dtype=np.uint16)
                k = len(cases)
                img["src%d" % k], img["dst%d" % k] = src, dst.copy()
                L.ref_init(IW, IH)
                L.ref_img_decoded(O._p(dst), O._p(full), O._p(c16), O._p(c16), O._p(src), cf, opa, recolor, recolor_opa, 0)
                L.ref_init(480, 480)
                img["out%d" % k] = dst
                cases.append([cf, recolor, recolor_opa, opa])
        img["cases"] = np.array(cases, np.int32)
        img["coords"] = c16
        img["chroma_key"] = np.array([key], np.int32)
        path = os.path.join(GOLD, "lvgl_image_recolor_swap%d.npz" % swap)
        np.savez_compressed(path, **img)
        print(path, os.path.getsize(path), "bytes")
        path = os.path.join(GOLD, "lvgl_sw_rotate_swap%d.npz" % swap)
        np.savez_compressed(path, **arrays)
        print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
