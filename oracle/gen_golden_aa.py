#!/usr/bin/env python3
"""Golden vectors for a display with anti-aliasing switched off (disp->driver->antialiasing == 0), from the REAL LVGL
(oracle/_ref, compiled from /root/reference): lv_draw_sw_blend_basic then rounds every mask to 0 / 255, in place,
before it blends (libraries/lvgl/src/draw/sw/lv_draw_sw_blend.c:150-156).  Writes tests/golden/aa_off_swap{0,1}.npz:
the expected draw buffers (and the masks as LVGL left them) of lv_draw_sw_blend over clipped rectangles for every
fill / map branch and of lv_draw_sw_img_decoded for a TRUE_COLOR_ALPHA image; the inputs are rebuilt by cases() from
the shared counter-based generator.
    python oracle/gen_golden_aa.py         (build container only)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import lvref_py as O  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
W, H = 67, 19


def cases():
    """Deterministic inputs: (name, kind, clip, blend_area, mask_area, src, color, mask, opa, mode, transp)."""
    out = []
    k = 0
    for transp in (0, 1):
        for has_src in (0, 1):
            for opa in (255, 253, 200, 3):
                for mode in ((0, 1, 2, 3) if (has_src or not transp) and opa == 200 else (0,)):
                    seed = 9000 + k
                    ba = (3 + k % 5, 1 + k % 3, 60 - k % 7, 17 - k % 4)
                    ma = (ba[0] - 1, ba[1], ba[2] + 2, ba[3] + 1)
                    clip = (5, 2, 58, 15)
                    bw, bh = ba[2] - ba[0] + 1, ba[3] - ba[1] + 1
                    mw, mh = ma[2] - ma[0] + 1, ma[3] - ma[1] + 1
                    src = O.synth(O.SYNTH_U16, seed, 0, bw * bh).reshape(bh, bw).copy() if has_src else None
                    mask = O.synth(O.SYNTH_MASK_UNI, seed + 1, 0, mw * mh).reshape(mh, mw).copy()
                    mask.reshape(-1)[::7] = 128      # the threshold itself (stays transparent) and its neighbours
                    mask.reshape(-1)[3::11] = 129
                    mask.reshape(-1)[5::13] = 127
                    color = int(O.synth(O.SYNTH_U16, seed + 2, 0, 1)[0])
                    out.append(dict(name="b%02d" % k, kind="blend", clip=clip, ba=ba, ma=ma, src=src, color=color, mask=mask,
                                    opa=opa, mode=mode, transp=transp, seed=seed))
                    k += 1
    for i, opa in enumerate((255, 180)):
        seed = 9500 + i
        co = (4, 2, 61, 16)
        cw, ch = co[2] - co[0] + 1, co[3] - co[1] + 1
        src = O.synth(O.SYNTH_MASK_UNI, seed, 0, cw * ch * 3).reshape(ch, cw, 3).copy()
        src[..., 2].reshape(-1)[::5] = 128
        out.append(dict(name="i%02d" % i, kind="img", clip=(6, 3, 59, 14), ba=co, src=src, opa=opa, seed=seed))
    return out


def dst_for(c):
    n = W * H * (3 if c.get("transp") else 1)
    kind = O.SYNTH_MASK_UNI if c.get("transp") else O.SYNTH_U16
    d = O.synth(kind, c["seed"] + 5, 0, n).copy()
    return d.reshape(H, W, 3) if c.get("transp") else d.reshape(H, W)


def main():
    full = (0, 0, W - 1, H - 1)
    for swap in (1, 0):
        L = O.ref(swap)
        if L is None:
            sys.exit("oracle/_ref is not built (make -C oracle ref)")
        L.ref_init(W, H)
        L.ref_set_antialiasing(0)
        arrays = {}
        for c in cases():
            d = dst_for(c)
            if c["kind"] == "blend":
                ba, ca, bla, ma = (O._area16(a) for a in (full, c["clip"], c["ba"], c["ma"]))
                m = c["mask"].copy()
                L.ref_blend(O._p(d), O._p(ba), O._p(ca), O._p(bla), O._p(c["src"]), c["color"], O._p(m), 2, O._p(ma), c["opa"],
                            c["mode"], c["transp"])
                arrays[c["name"] + "_mask"] = m
            else:
                ba, ca, co = (O._area16(a) for a in (full, c["clip"], c["ba"]))
                L.ref_img_decoded(O._p(d), O._p(ba), O._p(ca), O._p(co), O._p(c["src"]), 5, c["opa"], 0, 0, 0)
            arrays[c["name"]] = d
        L.ref_set_antialiasing(1)
        L.ref_init(480, 480)
        np.savez_compressed(os.path.join(GOLD, "aa_off_swap%d.npz" % swap), **arrays)
        print("swap %d: %d cases" % (swap, len(cases())))


if __name__ == "__main__":
    main()
