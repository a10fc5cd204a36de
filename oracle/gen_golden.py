#!/usr/bin/env python3
"""Generate tests/golden/*.npz + golden.json from the REAL LVGL (oracle/_ref, compiled from /root/reference).

Run in the build container only (`python oracle/gen_golden.py`).  Inputs are produced by the counter-based
generator shared by the oracle and the GPU library (so the files hold mostly expected OUTPUTS); every expected
value comes from LVGL's own code: lv_color_mix / lv_color_premult / lv_color_mix_premult / lv_color_hex /
lv_color_make / lv_color_to32 / lv_color_mix_with_alpha, lv_draw_sw_blend and lv_draw_sw_img_decoded.
The flush callback (src/lv_port.c) cannot be built here and has no golden vectors.
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import lvref_py as O  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
OPAS = [3, 127, 128, 251, 252, 253, 254, 255]
MIXES = [0, 3, 4, 127, 128, 251, 252, 255]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def synth_mask(seed, shape):
    """UI-like mask with extra mass on the threshold values."""
    n = int(np.prod(shape))
    m = O.synth(O.SYNTH_MASK_UI, seed, 0, n).copy()
    sel = O.synth(O.SYNTH_MASK_UNI, seed + 77, 0, n)
    m[sel < 12] = 252
    m[(sel >= 12) & (sel < 24)] = 253
    m[(sel >= 24) & (sel < 36)] = 254
    m[(sel >= 36) & (sel < 48)] = 1
    m[(sel >= 48) & (sel < 60)] = 2
    return m.reshape(shape)


def blend_cases():
    """(name, W, H, buf_area, clip, blend_area, mask_area|None, has_src, opa, mask_res, dst_kind, mode, transp)"""
    cases = []
    k = 0
    full = (0, 0, 319, 479)
    for has_src in (0, 1):
        for has_mask in (0, 1):
            for opa in OPAS:
                # clip != buffer, odd x1, partial rows, mask_area != blend_area
                cases.append(dict(name="c%03d" % k, W=320, H=480, buf_area=full, clip=(5, 7, 300, 401),
                                  blend_area=(3, 2, 311, 470), mask_area=(1, 0, 315, 475) if has_mask else None,
                                  has_src=has_src, opa=opa, mask_res=2, dst="rand", mode=0, transp=0))
                k += 1
    for opa in (3, 100, 128, 250, 252):  # A5(ii) cache cases
        for dst in ("lead_black", "black_after", "all_black"):
            cases.append(dict(name="c%03d" % k, W=320, H=480, buf_area=full, clip=(0, 0, 319, 479),
                              blend_area=(0, 0, 319, 479), mask_area=None, has_src=0, opa=opa, mask_res=2, dst=dst,
                              mode=0, transp=0))
            k += 1
            cases.append(dict(name="c%03d" % k, W=320, H=480, buf_area=full, clip=(10, 11, 200, 300),
                              blend_area=(-5, -5, 250, 400), mask_area=None, has_src=0, opa=opa, mask_res=2, dst=dst,
                              mode=0, transp=0))
            k += 1
    for mres in (0, 1, 3):  # mask_res handling
        cases.append(dict(name="c%03d" % k, W=64, H=40, buf_area=(10, 20, 73, 59), clip=(12, 22, 70, 50),
                          blend_area=(11, 21, 72, 58), mask_area=(11, 21, 72, 58), has_src=1, opa=200, mask_res=mres,
                          dst="rand", mode=0, transp=0))
        k += 1
    for mode in (1, 2, 3, 4):  # non-normal blend modes
        for has_src in (0, 1):
            for has_mask in (0, 1):
                for opa in (3, 128, 253, 255):
                    cases.append(dict(name="c%03d" % k, W=96, H=33, buf_area=(0, 0, 95, 32), clip=(1, 1, 90, 30),
                                      blend_area=(0, 0, 95, 32), mask_area=(0, 0, 95, 32) if has_mask else None,
                                      has_src=has_src, opa=opa, mask_res=2, dst="rand", mode=mode, transp=0))
                    k += 1
    for mode in (0, 1, 3):  # ARGB8565 destination (screen_transp)
        for has_src in (0, 1):
            for has_mask in (0, 1):
                for opa in (3, 128, 253, 254, 255):
                    if mode and not has_src:
                        continue
                    cases.append(dict(name="c%03d" % k, W=80, H=24, buf_area=(0, 0, 79, 23), clip=(2, 1, 77, 22),
                                      blend_area=(0, 0, 79, 23), mask_area=(0, 0, 79, 23) if has_mask else None,
                                      has_src=has_src, opa=opa, mask_res=2, dst="rand", mode=mode, transp=1))
                    k += 1
    return cases


def case_inputs(c, seed):
    """Deterministic inputs of a blend case (shared with the tests)."""
    W, H = c["W"], c["H"]
    if c["transp"]:
        dst = O.synth(O.SYNTH_MASK_UNI, seed, 0, W * H * 3).reshape(H, W, 3).copy()
        dst[..., 2] = synth_mask(seed + 5, (H, W))
    else:
        dst = O.synth(O.SYNTH_U16, seed, 0, W * H).reshape(H, W).copy()
        if c["dst"] == "lead_black":
            dst[: H // 2] = 0
            dst[H // 2, : W // 3] = 0
        elif c["dst"] == "black_after":
            dst[H // 2:] = 0
            dst[3, 5:90] = 0
        elif c["dst"] == "all_black":
            dst[:] = 0
    ba = c["blend_area"]
    bw, bh = ba[2] - ba[0] + 1, ba[3] - ba[1] + 1
    src = O.synth(O.SYNTH_U16, seed + 1, 0, bw * bh).reshape(bh, bw) if c["has_src"] else None
    mask = None
    if c["mask_area"] is not None:
        ma = c["mask_area"]
        mask = synth_mask(seed + 2, (ma[3] - ma[1] + 1, ma[2] - ma[0] + 1))
    color = int(O.synth(O.SYNTH_U16, seed + 3, 0, 1)[0])
    return dst, src, mask, color


def main():
    os.makedirs(GOLD, exist_ok=True)
    meta = {"generated_from": "LVGL 8.x sources of the reference, firmware lv_conf.h (swap1) / same options with "
                              "LV_COLOR_16_SWAP 0 (swap0); see oracle/Makefile", "swap": {}}
    for swap in (1, 0):
        L = O.ref(swap)
        assert L is not None, "oracle/_ref is not built (make -C oracle ref)"
        m = {}
        arrays = {}
        # ---- lv_color_mix: random triples + exhaustive digests ----
        n = 50000
        c1 = O.synth(O.SYNTH_U16, 11, 0, n)
        c2 = O.synth(O.SYNTH_U16, 12, 0, n)
        mix = O.synth(O.SYNTH_MASK_UNI, 13, 0, n)
        out = np.empty(n, np.uint16)
        L.ref_color_mix_n(O._p(c1), O._p(c2), O._p(mix), O._p(out), n)
        arrays["mix_out"] = out
        dig_full, dig_part = {}, {}
        for mx in MIXES:
            full = int(L.ref_color_mix_digest(mx, 0, 65536))
            mine = int(O.lib().lvref_color_mix_digest(mx, 0, 65536, swap))
            assert full == mine, "oracle lv_color_mix differs from LVGL at mix %d swap %d" % (mx, swap)
            dig_full[str(mx)] = full
            dig_part[str(mx)] = [int(L.ref_color_mix_digest(mx, 0, 1024)), int(L.ref_color_mix_digest(mx, 64512, 65536))]
        m["mix_digest_all_pairs"] = dig_full          # all 65536 x 65536 (fg, bg) pairs, validated at generation time
        m["mix_digest_fg_ranges"] = dig_part          # fg in [0,1024) and [64512,65536), all bg: cheap enough for tests
        # ---- premult / mix_premult ----
        pm = np.empty(3 * n, np.uint16)
        L.ref_color_premult_n(O._p(c1), O._p(mix), O._p(pm), n)
        arrays["premult_out"] = pm
        inv = O.synth(O.SYNTH_MASK_UNI, 14, 0, n)
        out2 = np.empty(n, np.uint16)
        L.ref_color_mix_premult_n(O._p(pm), O._p(c2), O._p(inv), O._p(out2), n)
        arrays["mix_premult_out"] = out2
        # ---- conversions: every input ----
        allhex = np.arange(1 << 24, dtype=np.uint32)
        o16 = np.empty(1 << 24, np.uint16)
        L.ref_color_hex_n(O._p(allhex), O._p(o16), 1 << 24)
        m["hex_all_sha256"] = sha(o16)
        r = (allhex >> 16).astype(np.uint8)
        g = ((allhex >> 8) & 0xFF).astype(np.uint8)
        b = (allhex & 0xFF).astype(np.uint8)
        o16b = np.empty(1 << 24, np.uint16)
        L.ref_color_make_n(O._p(r), O._p(g), O._p(b), O._p(o16b), 1 << 24)
        m["make_all_sha256"] = sha(o16b)
        all16 = np.arange(1 << 16, dtype=np.uint16)
        o32 = np.empty(1 << 16, np.uint32)
        L.ref_color_to32_n(O._p(all16), O._p(o32), 1 << 16)
        m["to32_all_sha256"] = sha(o32)
        arrays["to32_all"] = o32
        # ---- lv_color_mix_with_alpha ----
        bgo = O.synth(O.SYNTH_MASK_UNI, 15, 0, n)
        fgo = O.synth(O.SYNTH_MASK_UNI, 16, 0, n)
        res = np.empty(n, np.uint16)
        reso = np.empty(n, np.uint8)
        L.ref_color_mix_with_alpha_n(O._p(c2), O._p(bgo), O._p(c1), O._p(fgo), O._p(res), O._p(reso), n)
        arrays["mix_alpha_res"] = res
        arrays["mix_alpha_opa"] = reso
        # ---- blend cases ----
        cases = blend_cases()
        bl = {}
        for i, c in enumerate(cases):
            dst, src, mask, color = case_inputs(c, 1000 + 10 * i)
            O.ref_blend(L, dst, c["buf_area"], c["clip"], c["blend_area"], src=src, color=color, mask=mask,
                        mask_res=c["mask_res"], mask_area=c["mask_area"], opa=c["opa"], blend_mode=c["mode"],
                        screen_transp=c["transp"])
            bl[c["name"]] = sha(dst)
            if c["W"] * c["H"] <= 4096:
                arrays["blend_" + c["name"]] = dst
        m["blend_sha256"] = bl
        # ---- lv_draw_sw_img_decoded, LV_IMG_CF_TRUE_COLOR_ALPHA (3 bytes/px) ----
        img = {}
        for i, (W, H, clip, coords, opa) in enumerate([(320, 480, (0, 0, 319, 479), (0, 0, 319, 479), 255),
                                                        (320, 480, (20, 30, 250, 300), (10, 20, 259, 319), 255),
                                                        (320, 480, (20, 30, 250, 300), (10, 20, 259, 319), 180),
                                                        (100, 37, (0, 0, 99, 36), (0, 0, 99, 36), 253)]):
            iw, ih = coords[2] - coords[0] + 1, coords[3] - coords[1] + 1
            src = O.synth(O.SYNTH_MASK_UNI, 500 + i, 0, iw * ih * 3).reshape(ih, iw, 3).copy()
            src[..., 2] = synth_mask(600 + i, (ih, iw))
            dst = O.synth(O.SYNTH_U16, 700 + i, 0, W * H).reshape(H, W).copy()
            L.ref_init(W, H)
            area = np.array([0, 0, W - 1, H - 1], np.int16)
            L.ref_img_decoded(O._p(dst), O._p(area), O._p(np.array(clip, np.int16)), O._p(np.array(coords, np.int16)),
                              O._p(src), 5, opa, 0, 0, 0)
            img["i%d" % i] = dict(W=W, H=H, clip=clip, coords=coords, opa=opa, sha256=sha(dst))
        L.ref_init(480, 480)
        m["img_decoded"] = img
        # ---- layout of the boundary types ----
        keys = ["sizeof(lv_area_t)", "sizeof(lv_color_t)", "sizeof(lv_draw_sw_blend_dsc_t)", "blend_dsc.blend_area",
                "blend_dsc.src_buf", "blend_dsc.color", "blend_dsc.mask_buf", "blend_dsc.mask_res",
                "blend_dsc.mask_area", "blend_dsc.opa", "blend_dsc.blend_mode", "sizeof(lv_draw_ctx_t)",
                "draw_ctx.buf", "draw_ctx.buf_area", "draw_ctx.clip_area", "draw_ctx.init_buf",
                "draw_ctx.wait_for_finish", "draw_ctx.buffer_copy", "draw_ctx.layer_instance_size",
                "draw_ctx.user_data", "sizeof(lv_draw_sw_ctx_t)", "sw_ctx.blend", "sizeof(lv_disp_draw_buf_t)",
                "draw_buf.buf1", "draw_buf.buf_act", "draw_buf.size", "draw_buf.flushing", "draw_buf.flushing_last",
                "sizeof(lv_disp_drv_t)", "disp_drv.hor_res", "disp_drv.ver_res", "disp_drv.draw_buf",
                "disp_drv.flush_cb", "disp_drv.set_px_cb", "disp_drv.draw_ctx", "disp_drv.draw_ctx_init",
                "disp_drv.draw_ctx_deinit", "disp_drv.draw_ctx_size", "disp_drv.user_data",
                "LV_DRAW_MASK_RES_TRANSP", "LV_DRAW_MASK_RES_FULL_COVER", "LV_DRAW_MASK_RES_CHANGED",
                "LV_BLEND_MODE_NORMAL", "LV_BLEND_MODE_ADDITIVE", "LV_BLEND_MODE_SUBTRACTIVE", "LV_BLEND_MODE_MULTIPLY",
                "LV_DISP_ROT_90", "LV_DISP_ROT_180", "LV_DISP_ROT_270", "LV_IMG_CF_TRUE_COLOR_ALPHA"]
        m["layout"] = {k: int(L.ref_layout(k.encode())) for k in keys}
        m["disp_drv_bits"] = {k: int(L.ref_disp_drv_bits(k.encode())) for k in
                              ("direct_mode", "full_refresh", "sw_rotate", "antialiasing", "rotated", "screen_transp")}
        np.savez_compressed(os.path.join(GOLD, "swap%d.npz" % swap), **arrays)
        meta["swap"][str(swap)] = m
        print("swap", swap, "done:", len(cases), "blend cases")
    json.dump(meta, open(os.path.join(GOLD, "golden.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
