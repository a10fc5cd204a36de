"""CPU: the C-ABI library loads, exports every function its headers declare, and its PODs have the layout the
headers (and LVGL's own types) say.  No compute calls here."""
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
INC = os.path.join(ROOT, "include")


def test_library_exports_every_declared_symbol():
    from vvc_amd import _lib
    L = _lib.lib()
    names = _lib.declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(L, n), "liblvgpu.so does not export %s" % n
    assert L.lvgpu_version() == 0x000100


def test_argument_errors_need_no_gpu():
    from vvc_amd import _lib
    L = _lib.lib()
    assert L.lvgpu_blend_batch(None, None) == -1
    assert b"NULL" in L.lvgpu_last_error()
    assert L.lvgpu_flush_rotate_batch(None, None) == -1
    f = _lib.FlushBatch()
    f.width, f.height, f.rotation, f.n_frames = 8, 8, 7, 1
    assert L.lvgpu_flush_rotate_batch(C.byref(f), None) == -1
    b = _lib.BlendBatch()
    b.n_frames = -1
    assert L.lvgpu_blend_batch(C.byref(b), None) == -1
    b.n_frames = 0
    assert L.lvgpu_blend_batch(C.byref(b), None) == 0        # empty batch is a no-op
    assert L.lvgpu_color_fill(None, 0, 0, None) == 0
    assert L.lvgpu_color_fill(None, 0, 4, None) == -1


PROBE = r"""
#include <stdio.h>
#include <stddef.h>
#include "lvgpu.h"
#include "lvgpu_lvgl.h"
#define P(name, v) printf("\"%s\": %ld,\n", name, (long)(v))
int main(void) {
    printf("{\n");
    P("sizeof(lvgpu_area_t)", sizeof(lvgpu_area_t));
    P("sizeof(lvgpu_blend_batch_t)", sizeof(lvgpu_blend_batch_t));
    P("blend.dst", offsetof(lvgpu_blend_batch_t, dst));
    P("blend.buf_area", offsetof(lvgpu_blend_batch_t, buf_area));
    P("blend.src", offsetof(lvgpu_blend_batch_t, src));
    P("blend.mask", offsetof(lvgpu_blend_batch_t, mask));
    P("blend.mask_area", offsetof(lvgpu_blend_batch_t, mask_area));
    P("blend.n_frames", offsetof(lvgpu_blend_batch_t, n_frames));
    P("blend.color", offsetof(lvgpu_blend_batch_t, color));
    P("blend.color_16_swap", offsetof(lvgpu_blend_batch_t, color_16_swap));
    P("sizeof(lvgpu_flush_batch_t)", sizeof(lvgpu_flush_batch_t));
    P("flush.out", offsetof(lvgpu_flush_batch_t, out));
    P("flush.width", offsetof(lvgpu_flush_batch_t, width));
    P("flush.n_frames", offsetof(lvgpu_flush_batch_t, n_frames));
    P("sizeof(lvgpu_blend_flush_batch_t)", sizeof(lvgpu_blend_flush_batch_t));
    P("bf.out", offsetof(lvgpu_blend_flush_batch_t, out));
    P("bf.rotation", offsetof(lvgpu_blend_flush_batch_t, rotation));
    /* mirrors of the LVGL types */
    P("sizeof(lv_area_t)", sizeof(lvgpu_lv_area_t));
    P("sizeof(lv_color_t)", sizeof(lvgpu_lv_color_t));
    P("sizeof(lv_draw_sw_blend_dsc_t)", sizeof(lvgpu_lv_draw_sw_blend_dsc_t));
    P("blend_dsc.blend_area", offsetof(lvgpu_lv_draw_sw_blend_dsc_t, blend_area));
    P("blend_dsc.src_buf", offsetof(lvgpu_lv_draw_sw_blend_dsc_t, src_buf));
    P("blend_dsc.color", offsetof(lvgpu_lv_draw_sw_blend_dsc_t, color));
    P("blend_dsc.mask_buf", offsetof(lvgpu_lv_draw_sw_blend_dsc_t, mask_buf));
    P("blend_dsc.mask_res", offsetof(lvgpu_lv_draw_sw_blend_dsc_t, mask_res));
    P("blend_dsc.mask_area", offsetof(lvgpu_lv_draw_sw_blend_dsc_t, mask_area));
    P("blend_dsc.opa", offsetof(lvgpu_lv_draw_sw_blend_dsc_t, opa));
    P("blend_dsc.blend_mode", offsetof(lvgpu_lv_draw_sw_blend_dsc_t, blend_mode));
    P("sizeof(lv_draw_ctx_t)", sizeof(lvgpu_lv_draw_ctx_t));
    P("draw_ctx.buf", offsetof(lvgpu_lv_draw_ctx_t, buf));
    P("draw_ctx.buf_area", offsetof(lvgpu_lv_draw_ctx_t, buf_area));
    P("draw_ctx.clip_area", offsetof(lvgpu_lv_draw_ctx_t, clip_area));
    P("draw_ctx.init_buf", offsetof(lvgpu_lv_draw_ctx_t, init_buf));
    P("draw_ctx.wait_for_finish", offsetof(lvgpu_lv_draw_ctx_t, wait_for_finish));
    P("draw_ctx.buffer_copy", offsetof(lvgpu_lv_draw_ctx_t, buffer_copy));
    P("draw_ctx.layer_instance_size", offsetof(lvgpu_lv_draw_ctx_t, layer_instance_size));
    P("draw_ctx.user_data", offsetof(lvgpu_lv_draw_ctx_t, user_data));
    P("sizeof(lv_disp_draw_buf_t)", sizeof(lvgpu_lv_disp_draw_buf_t));
    P("draw_buf.buf1", offsetof(lvgpu_lv_disp_draw_buf_t, buf1));
    P("draw_buf.buf_act", offsetof(lvgpu_lv_disp_draw_buf_t, buf_act));
    P("draw_buf.size", offsetof(lvgpu_lv_disp_draw_buf_t, size));
    P("draw_buf.flushing", offsetof(lvgpu_lv_disp_draw_buf_t, flushing));
    P("draw_buf.flushing_last", offsetof(lvgpu_lv_disp_draw_buf_t, flushing_last));
    P("sizeof(lv_disp_drv_t)", sizeof(lvgpu_lv_disp_drv_t));
    P("disp_drv.hor_res", offsetof(lvgpu_lv_disp_drv_t, hor_res));
    P("disp_drv.ver_res", offsetof(lvgpu_lv_disp_drv_t, ver_res));
    P("disp_drv.draw_buf", offsetof(lvgpu_lv_disp_drv_t, draw_buf));
    P("disp_drv.flush_cb", offsetof(lvgpu_lv_disp_drv_t, flush_cb));
    P("disp_drv.set_px_cb", offsetof(lvgpu_lv_disp_drv_t, set_px_cb));
    P("disp_drv.draw_ctx", offsetof(lvgpu_lv_disp_drv_t, draw_ctx));
    P("disp_drv.draw_ctx_init", offsetof(lvgpu_lv_disp_drv_t, draw_ctx_init));
    P("disp_drv.draw_ctx_deinit", offsetof(lvgpu_lv_disp_drv_t, draw_ctx_deinit));
    P("disp_drv.draw_ctx_size", offsetof(lvgpu_lv_disp_drv_t, draw_ctx_size));
    P("disp_drv.user_data", offsetof(lvgpu_lv_disp_drv_t, user_data));
    /* bitfield word: set each field in a zeroed struct and print the word at offset 24 */
    {
        const char *names[] = {"direct_mode", "full_refresh", "sw_rotate", "antialiasing", "rotated", "screen_transp"};
        for(int i = 0; i < 6; i++) {
            lvgpu_lv_disp_drv_t d; unsigned w;
            __builtin_memset(&d, 0, sizeof d);
            if(i == 0) d.direct_mode = 1; if(i == 1) d.full_refresh = 1; if(i == 2) d.sw_rotate = 1;
            if(i == 3) d.antialiasing = 1; if(i == 4) d.rotated = 3; if(i == 5) d.screen_transp = 1;
            __builtin_memcpy(&w, (char *)&d + 24, 4);
            printf("\"bits.%s\": %u,\n", names[i], w);
        }
    }
    P("LVGPU_MASK_RES_CHANGED", LVGPU_MASK_RES_CHANGED);
    printf("\"end\": 0}\n");
    return 0;
}
"""


@pytest.fixture(scope="module")
def probe():
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "probe.c")
        open(src, "w").write(PROBE)
        exe = os.path.join(td, "probe")
        subprocess.check_call(["gcc", "-std=gnu11", "-Wall", "-I", INC, src, "-o", exe])
        return json.loads(subprocess.check_output([exe]).decode())


def test_pod_layout_matches_ctypes(probe):
    from vvc_amd import _lib
    assert probe["sizeof(lvgpu_area_t)"] == C.sizeof(_lib.Area) == 8
    assert probe["sizeof(lvgpu_blend_batch_t)"] == C.sizeof(_lib.BlendBatch)
    for f in ("dst", "buf_area", "src", "mask", "mask_area", "n_frames", "color", "color_16_swap"):
        assert probe["blend." + f] == getattr(_lib.BlendBatch, f).offset, f
    assert probe["sizeof(lvgpu_flush_batch_t)"] == C.sizeof(_lib.FlushBatch)
    assert probe["flush.out"] == _lib.FlushBatch.out.offset
    assert probe["flush.width"] == _lib.FlushBatch.width.offset
    assert probe["flush.n_frames"] == _lib.FlushBatch.n_frames.offset
    assert probe["sizeof(lvgpu_blend_flush_batch_t)"] == C.sizeof(_lib.BlendFlushBatch)
    assert probe["bf.out"] == _lib.BlendFlushBatch.out.offset
    assert probe["bf.rotation"] == _lib.BlendFlushBatch.rotation.offset


def test_lvgl_mirror_layout_matches_real_lvgl(probe):
    """The mirror structs of include/lvgpu_lvgl.h against offsetof/sizeof taken from the real LVGL headers
    (recorded in tests/golden/golden.json by oracle/gen_golden.py)."""
    meta = json.load(open(os.path.join(HERE, "golden", "golden.json")))
    for swap in ("0", "1"):
        lay = meta["swap"][swap]["layout"]
        for k, v in lay.items():
            if k in probe:
                assert probe[k] == v, (k, probe[k], v)
        for k, v in meta["swap"][swap]["disp_drv_bits"].items():
            assert probe["bits." + k] == v, k
        assert lay["LV_DRAW_MASK_RES_CHANGED"] == probe["LVGPU_MASK_RES_CHANGED"] == 2
        assert (lay["LV_BLEND_MODE_NORMAL"], lay["LV_BLEND_MODE_ADDITIVE"], lay["LV_BLEND_MODE_SUBTRACTIVE"],
                lay["LV_BLEND_MODE_MULTIPLY"]) == (0, 1, 2, 3)
        assert (lay["LV_DISP_ROT_90"], lay["LV_DISP_ROT_180"], lay["LV_DISP_ROT_270"]) == (1, 2, 3)
        assert lay["LV_IMG_CF_TRUE_COLOR_ALPHA"] == 5


def test_draw_ctx_size_covers_lv_draw_sw_ctx():
    from vvc_amd import _lib
    meta = json.load(open(os.path.join(HERE, "golden", "golden.json")))
    assert _lib.lib().lvgpu_draw_ctx_size() >= meta["swap"]["1"]["layout"]["sizeof(lv_draw_sw_ctx_t)"]


def test_product_does_not_import_the_oracle():
    """The oracle is test infrastructure: nothing under vvc_amd/ may mention it."""
    for dp, _, fs in os.walk(os.path.join(ROOT, "vvc_amd")):
        for f in fs:
            if f.endswith((".py", ".hip", ".cpp", ".hpp", ".h")):
                text = open(os.path.join(dp, f), errors="ignore").read()
                assert "lvref" not in text and "oracle" not in text.lower().replace("oracle/_ref", ""), os.path.join(dp, f)


def test_flush_chunks_match_the_restated_flush_callback(oracle):
    """lvgpu_flush_chunks (host arithmetic of the C ABI) against the oracle's literal restatement of the reference's
    chunk loop (src/lv_port.c:466-566): same rectangles in the same order, and every chunk is the stated contiguous
    slice of the area's panel image."""
    import numpy as np
    from vvc_amd import lvgpu
    rng = np.random.default_rng(12)
    for (W, H, trans) in [(480, 320, 480 * 320 // 10), (320, 480, 15360), (100, 37, 1000), (64, 64, 64 * 64), (33, 7, 40),
                          (480, 320, 0x7FFFFFFF)]:
        frame = rng.integers(0, 65536, (H, W), dtype=np.uint16)
        for rot in (0, 1, 2, 3):
            exp = oracle.flush((0, 0, W - 1, H - 1), frame, rot, trans, W, H)
            panel = oracle.flush_frame(frame, W, H, rot, 0)
            got = lvgpu.flush_chunks((0, 0, W - 1, H - 1), W, H, rot, trans)
            assert len(got) == len(exp), (W, H, rot)
            for g, e in zip(got, exp):
                assert g[:4] == e[:4], (W, H, rot, g, e[:4])
                assert g[5] == (g[2] - g[0]) * (g[3] - g[1])
                assert np.array_equal(panel[g[4]:g[4] + g[5]], e[4])
    with pytest.raises(lvgpu.LvgpuError):
        lvgpu.flush_chunks((0, 0, 99, 9), 100, 10, 1, 5)      # trans_size below one line
    with pytest.raises(lvgpu.LvgpuError):
        lvgpu.flush_chunks((0, 0, 99, 9), 100, 10, 7, 500)    # bad rotation


def test_panel_transactions_match_the_restated_panel_driver(oracle):
    """lvgpu_axs15231b_transactions against the oracle's restatement of panel_axs15231b_draw_bitmap
    (src/esp_lcd_axs15231b.c:292-337; parity unpinned -- the driver needs ESP-IDF headers the reference tree does not
    hold) for every transport chunk of a flushed frame, plus the known answer for the firmware's geometry: ten QSPI
    chunks of 48 panel rows, column window 0..319, RAMWR for the first and RAMWRC for the rest."""
    from vvc_amd import lvgpu
    for rot in (0, 1, 2, 3):
        for (xg, yg, qspi, bpp) in [(0, 0, True, 16), (0, 0, False, 16), (3, 5, True, 16), (2, 0, False, 24)]:
            for ch in lvgpu.flush_chunks((0, 0, 479, 319), 480, 320, rot, 15360):
                got = lvgpu.axs15231b_transactions(ch, xg, yg, qspi, bpp)
                exp = oracle.axs15231b_draw_bitmap(ch[0], ch[1], ch[2], ch[3], xg, yg, qspi, bpp)
                assert len(got) == len(exp) == (2 if qspi else 3)
                for g, e in zip(got, exp):
                    assert (g[0], g[1], g[2]) == (e[0], e[1], e[2]), (rot, ch, g, e)
                    if g[0]:
                        assert g[3] == ch[4] and g[4] == e[3] == ch[5] * bpp // 8
    chunks = lvgpu.flush_chunks((0, 0, 479, 319), 480, 320, 1, 15360)
    assert len(chunks) == 10
    for k, ch in enumerate(chunks):
        t = lvgpu.axs15231b_transactions(ch)
        assert t[0] == (0, 0x02002A00, bytes([0, 0, 0x01, 0x3F]), 0, 0)
        assert t[1] == (1, 0x32002C00 if k == 0 else 0x32003C00, b"", k * 15360, 30720)
    with pytest.raises(lvgpu.LvgpuError):
        lvgpu.axs15231b_transactions((5, 0, 5, 10, 0, 0))      # empty rectangle: the driver asserts
    with pytest.raises(lvgpu.LvgpuError):
        lvgpu.axs15231b_transactions(chunks[0], bits_per_pixel=12)


def test_panel_transactions_rebuild_the_panel_image(oracle):
    """The framing consumed: the oracle's panel image of a flushed frame, cut into the transport chunks and framed as
    the driver frames them, fed to a model of the panel's frame memory (tests/util_panel.py), gives that image back --
    RAMWR once, RAMWRC for every later chunk, every pixel written exactly once.  (The same walk over a panel image
    that a fused GPU launch produced is tests/test_gpu_wire.py.)  Parity unpinned, as above."""
    import numpy as np
    from util_panel import RAMWR, RAMWRC, PanelModel, send
    from vvc_amd import lvgpu
    rng = np.random.default_rng(3)
    for rot, W, H, trans in [(1, 480, 320, 15360), (0, 320, 480, 15360), (3, 480, 320, 15360), (2, 320, 480, 15360),
                             (1, 200, 96, 1000), (2, 100, 37, 450)]:
        frame = rng.integers(0, 65536, (H, W), dtype=np.uint16)
        img = oracle.flush_frame(frame, W, H, rot, 0)
        hor, ver = (H, W) if rot in (1, 3) else (W, H)
        for qspi in (True, False):
            m = PanelModel(hor, ver, qspi)
            chunks = send(lvgpu, img.tobytes(), (0, 0, W - 1, H - 1), W, H, rot, trans, m)
            per = 2 if qspi else 3
            assert m.log[per - 1] == RAMWR and all(c == RAMWRC for c in m.log[2 * per - 1::per])
            assert len(chunks) > 1 and (m.written == 1).all()
            assert m.gram.tobytes() == img.tobytes(), (rot, W, H, qspi)


def _plan_of_scene(path, W=480, H=320, flush=1):
    """lvgpu_blend_list_plan on a golden display list; the pointers are placeholders (the plan never touches memory)."""
    from util import load_scene
    from vvc_amd import _lib
    fb, items = load_scene(path)
    arr = (_lib.BlendBatch * len(items))()
    total_px = 0
    for i, it in enumerate(items):
        b = arr[i]
        b.dst = 0x10000
        b.dst_frame_stride = W * H
        b.buf_area = _lib.Area(0, 0, W - 1, H - 1)
        b.clip_area = b.blend_area = b.mask_area = _lib.Area(*it["area"])
        b.n_frames = 4
        b.src_format = 0 if it["src"] is None else 1
        b.src = None if it["src"] is None else 0x20000
        b.mask = None if it["mask"] is None else 0x30000
        b.mask_res = 2
        b.color, b.opa, b.blend_mode, b.color_16_swap = it["color"], it["opa"], it["blend_mode"], 1
        if it["opa"] > 2:
            total_px += (it["area"][2] - it["area"][0] + 1) * (it["area"][3] - it["area"][1] + 1)
    plan = _lib.ListPlan()
    rc = _lib.lib().lvgpu_blend_list_plan(None, arr, len(items), flush, 1, C.byref(plan))
    assert rc == 0, _lib.lib().lvgpu_last_error()
    return plan.as_dict(), len(items), total_px


def test_list_plan_of_recorded_lvgl_scenes():
    """Host side of the display lists (binning, batching, occlusion culling) on the golden LVGL recordings, no GPU:
    every clipped pixel of every listed item is accounted for exactly once, the hundreds of one-row descriptors
    collapse into few steps per tile, and the opaque screen background makes every tile write-only."""
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    for name, n_exp in (("scene0_f0_swap1.npz", 762), ("scene1_f0_swap1.npz", 888), ("scene2_f0_swap1.npz", 1649)):
        p, n, total_px = _plan_of_scene(os.path.join(gold, name))
        assert n == n_exp
        assert p["items_listed"] + p["items_standalone"] + p["items_noop"] == n
        assert p["items_standalone"] == 0 and p["launches"] == 1 and p["flush_fused"] == 1
        assert p["tiles"] == 8 * 5 and p["tiles_selected"] == 40 and p["tiles_overwritten"] == 40
        assert p["px_item_steps"] + p["px_chunks"] + p["px_culled"] == total_px
        assert p["px_culled"] == 0  # nothing but the background covers a whole tile opaquely in these scenes
        steps = p["item_steps"] + p["batch_steps"]
        assert steps < n // 3, "one-row descriptors must be batched (%d steps for %d items)" % (steps, n)
        assert p["chunks_full"] > 4 * p["chunks_partial"]
    # without the flush only tiles with work are visited (all of them here: the background touches every tile)
    p2, _, _ = _plan_of_scene(os.path.join(gold, "scene0_f0_swap1.npz"), flush=0)
    assert p2["flush_fused"] == 0 and p2["launches"] == 1 and p2["tiles_selected"] == 40


def test_list_plan_culling_and_fallbacks():
    from vvc_amd import _lib

    def mk(area, color=1, opa=255, W=128, H=128, dst=0x10000, clip=None):
        b = _lib.BlendBatch()
        b.dst, b.dst_frame_stride, b.n_frames = dst, W * H, 2
        b.buf_area = _lib.Area(0, 0, W - 1, H - 1)
        b.clip_area = b.blend_area = b.mask_area = _lib.Area(*area)
        if clip is not None:
            b.clip_area = _lib.Area(*clip)
        b.color, b.opa, b.color_16_swap, b.mask_res = color, opa, 1, 2
        return b

    def plan(items, flush=0):
        arr = (_lib.BlendBatch * len(items))(*items)
        p = _lib.ListPlan()
        assert _lib.lib().lvgpu_blend_list_plan(None, arr, len(items), flush, 0, C.byref(p)) == 0
        return p.as_dict()

    full = (0, 0, 127, 127)
    # three opaque full-screen fills: only the last survives in every tile
    p = plan([mk(full, 1), mk(full, 2), mk(full, 3)])
    assert p["px_culled"] == 2 * 128 * 128 and p["px_item_steps"] == 128 * 128 and p["tiles_overwritten"] == 4
    # a translucent fill whose value on black depends on scan order leaves the tile pass: 3 launches
    # (stored colour 0x00F0 = native red 30 under LV_COLOR_16_SWAP 1; opa 135: lv_color_mix gives 15 on black, the
    # premultiplied formula 16)
    p = plan([mk(full, 1), mk((3, 3, 40, 40), 0x00F0, opa=135), mk((0, 0, 9, 0), 5)])
    assert (p["items_listed"], p["items_standalone"], p["launches"]) == (2, 1, 3)
    # an unaligned buffer cannot use the tile pass at all: every item on its own
    p = plan([mk(full, 1, dst=0x10002), mk((1, 1, 5, 5), 2, dst=0x10002)])
    assert (p["items_listed"], p["items_standalone"], p["launches"]) == (0, 2, 2)
    # opa <= LV_OPA_MIN and empty clips are dropped
    p = plan([mk(full, 1, opa=2), mk((200, 200, 210, 210), 1, clip=full)])
    assert p["items_noop"] == 2 and p["launches"] == 0
    # a clip that reaches outside the buffer is an error, as in lvgpu_blend_batch
    arr = (_lib.BlendBatch * 1)(mk((120, 120, 130, 130), 1))
    assert _lib.lib().lvgpu_blend_list_plan(None, arr, 1, 0, 0, C.byref(_lib.ListPlan())) == -1
