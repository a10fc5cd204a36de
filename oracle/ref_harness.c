/*
 * ref_harness.c -- TEST INFRASTRUCTURE ONLY.
 *
 * A thin C entry layer (this repository's own code) that is compiled TOGETHER WITH the real, unmodified LVGL 8.x
 * sources of the reference (libraries/lvgl/src, found under /root/reference at build time; see oracle/Makefile)
 * into oracle/_ref/liblvgl_swap{0,1}.so.  It lets the tests, the golden-vector generator and bench.py's
 * cpu_baseline leg drive LVGL's own CPU renderer headlessly through plain pointers (ctypes friendly).
 * Nothing in the product path links or loads it.
 *
 * Headless drive follows the reference's own call order: lv_init -> lv_disp_draw_buf_init -> lv_disp_drv_init ->
 * lv_disp_drv_register (libraries/lvgl/src/hal/lv_hal_disp.c:84-243), then _lv_refr_set_disp_refreshing
 * (libraries/lvgl/src/core/lv_refr.c:274-277) because lv_draw_sw_blend_basic dereferences the refreshing display
 * (libraries/lvgl/src/draw/sw/lv_draw_sw_blend.c:122).
 */
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "lvgl.h"
#include "src/draw/sw/lv_draw_sw.h"
#include "src/core/lv_refr.h"

#define REF_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------------------------ */
/* Display 0: a bare display used only to host direct lv_draw_sw_blend / lv_draw_sw_img_decoded calls.           */
/* ------------------------------------------------------------------------------------------------------------ */

static int g_inited;
static lv_disp_draw_buf_t g_dbuf0;
static lv_disp_drv_t g_drv0;
static lv_disp_t * g_disp0;
static lv_color_t * g_buf0;

static void null_flush(lv_disp_drv_t * drv, const lv_area_t * a, lv_color_t * p)
{
    (void)a;
    (void)p;
    lv_disp_flush_ready(drv);
}

REF_API int ref_color_16_swap(void)
{
    return LV_COLOR_16_SWAP;
}

REF_API int ref_init(int hor_res, int ver_res)
{
    if(!g_inited) {
        lv_init();
        g_inited = 1;
    }
    if(g_disp0 == NULL) {
        g_buf0 = malloc((size_t)hor_res * ver_res * sizeof(lv_color_t));
        lv_disp_draw_buf_init(&g_dbuf0, g_buf0, NULL, (uint32_t)hor_res * ver_res);
        lv_disp_drv_init(&g_drv0);
        g_drv0.hor_res = (lv_coord_t)hor_res;
        g_drv0.ver_res = (lv_coord_t)ver_res;
        g_drv0.flush_cb = null_flush;
        g_drv0.draw_buf = &g_dbuf0;
        g_disp0 = lv_disp_drv_register(&g_drv0);
    }
    else {
        g_drv0.hor_res = (lv_coord_t)hor_res;
        g_drv0.ver_res = (lv_coord_t)ver_res;
    }
    _lv_refr_set_disp_refreshing(g_disp0);
    return g_disp0 ? 0 : -1;
}

static void set_area(lv_area_t * a, const int16_t * v)
{
    a->x1 = v[0];
    a->y1 = v[1];
    a->x2 = v[2];
    a->y2 = v[3];
}

/* One call of lv_draw_sw_blend (libraries/lvgl/src/draw/sw/lv_draw_sw_blend.c:95) on caller memory.
 * Areas are int16 {x1,y1,x2,y2}, inclusive, absolute.  screen_transp selects the ARGB8565 destination layout. */
REF_API void ref_blend(void * buf, const int16_t * buf_area, const int16_t * clip_area, const int16_t * blend_area,
                       const uint16_t * src_buf, uint16_t color, uint8_t * mask_buf, int mask_res,
                       const int16_t * mask_area, int opa, int blend_mode, int screen_transp)
{
    lv_area_t ba, ca, bla, ma;
    set_area(&ba, buf_area);
    set_area(&ca, clip_area);
    set_area(&bla, blend_area);
    if(mask_area) set_area(&ma, mask_area);

    _lv_refr_set_disp_refreshing(g_disp0);
    g_drv0.screen_transp = screen_transp ? 1 : 0;

    lv_draw_ctx_t * ctx = g_drv0.draw_ctx;
    ctx->buf = buf;
    ctx->buf_area = &ba;
    ctx->clip_area = &ca;

    lv_draw_sw_blend_dsc_t dsc;
    memset(&dsc, 0, sizeof(dsc));
    dsc.blend_area = &bla;
    dsc.src_buf = (const lv_color_t *)src_buf;
    dsc.color.full = color;
    dsc.mask_buf = mask_buf;
    dsc.mask_res = (lv_draw_mask_res_t)mask_res;
    dsc.mask_area = mask_area ? &ma : NULL;
    dsc.opa = (lv_opa_t)opa;
    dsc.blend_mode = (lv_blend_mode_t)blend_mode;
    lv_draw_sw_blend(ctx, &dsc);

    g_drv0.screen_transp = 0;
}

/* disp->driver->antialiasing of the display ref_blend and ref_img_decoded draw on (default 1,
 * libraries/lvgl/src/hal/lv_hal_disp.c:94); 0 makes lv_draw_sw_blend_basic round every mask in place
 * (libraries/lvgl/src/draw/sw/lv_draw_sw_blend.c:150-156). */
REF_API void ref_set_antialiasing(int on) { g_drv0.antialiasing = on ? 1 : 0; }

/* lv_draw_sw_img_decoded (libraries/lvgl/src/draw/sw/lv_draw_sw_img.c:44): exercises convert_cb (:203-298)
 * and the <= hor_res pixel chunking (:109-191) in front of the blend. */
REF_API void ref_img_decoded(void * buf, const int16_t * buf_area, const int16_t * clip_area, const int16_t * coords,
                             const uint8_t * src, int cf, int opa, uint16_t recolor, int recolor_opa, int blend_mode)
{
    lv_area_t ba, ca, co;
    set_area(&ba, buf_area);
    set_area(&ca, clip_area);
    set_area(&co, coords);

    _lv_refr_set_disp_refreshing(g_disp0);
    lv_draw_ctx_t * ctx = g_drv0.draw_ctx;
    ctx->buf = buf;
    ctx->buf_area = &ba;
    ctx->clip_area = &ca;

    lv_draw_img_dsc_t d;
    lv_draw_img_dsc_init(&d);
    d.opa = (lv_opa_t)opa;
    d.recolor.full = recolor;
    d.recolor_opa = (lv_opa_t)recolor_opa;
    d.blend_mode = (lv_blend_mode_t)blend_mode;
    lv_draw_sw_img_decoded(ctx, &d, &co, src, (lv_img_cf_t)cf);
}

REF_API void ref_buffer_copy(void * dest_buf, int dest_stride, const int16_t * dest_area, void * src_buf,
                             int src_stride, const int16_t * src_area)
{
    lv_area_t da, sa;
    set_area(&da, dest_area);
    set_area(&sa, src_area);
    lv_draw_sw_buffer_copy(g_drv0.draw_ctx, dest_buf, (lv_coord_t)dest_stride, &da, src_buf, (lv_coord_t)src_stride,
                           &sa);
}

/* ------------------------------------------------------------------------------------------------------------ */
/* Colour arithmetic, bulk forms (libraries/lvgl/src/misc/lv_color.h).                                          */
/* ------------------------------------------------------------------------------------------------------------ */

REF_API void ref_color_mix_n(const uint16_t * c1, const uint16_t * c2, const uint8_t * mix, uint16_t * out, size_t n)
{
    for(size_t i = 0; i < n; i++) {
        lv_color_t a, b;
        a.full = c1[i];
        b.full = c2[i];
        out[i] = lv_color_mix(a, b, mix[i]).full;
    }
}

/* Exhaustive helper: digest input is every (fg,bg) pair for one mix value; out must hold 65536*65536 entries if
 * used naively, so instead fold into a 64-bit FNV-1a here. */
REF_API uint64_t ref_color_mix_digest(int mix, uint32_t fg_lo, uint32_t fg_hi)
{
    uint64_t h = 1469598103934665603ULL;
    for(uint32_t fg = fg_lo; fg < fg_hi; fg++) {
        for(uint32_t bg = 0; bg < 65536; bg++) {
            lv_color_t a, b;
            a.full = (uint16_t)fg;
            b.full = (uint16_t)bg;
            uint16_t r = lv_color_mix(a, b, (uint8_t)mix).full;
            h = (h ^ (r & 0xFF)) * 1099511628211ULL;
            h = (h ^ (r >> 8)) * 1099511628211ULL;
        }
    }
    return h;
}

REF_API void ref_color_premult_n(const uint16_t * c, const uint8_t * mix, uint16_t * out3, size_t n)
{
    for(size_t i = 0; i < n; i++) {
        lv_color_t a;
        a.full = c[i];
        lv_color_premult(a, mix[i], out3 + 3 * i);
    }
}

REF_API void ref_color_mix_premult_n(const uint16_t * pm3, const uint16_t * c2, const uint8_t * mix, uint16_t * out,
                                     size_t n)
{
    for(size_t i = 0; i < n; i++) {
        uint16_t pm[3] = {pm3[3 * i], pm3[3 * i + 1], pm3[3 * i + 2]};
        lv_color_t b;
        b.full = c2[i];
        out[i] = lv_color_mix_premult(pm, b, mix[i]).full;
    }
}

REF_API void ref_color_hex_n(const uint32_t * c, uint16_t * out, size_t n)
{
    for(size_t i = 0; i < n; i++) out[i] = lv_color_hex(c[i]).full;
}

REF_API void ref_color_make_n(const uint8_t * r, const uint8_t * g, const uint8_t * b, uint16_t * out, size_t n)
{
    for(size_t i = 0; i < n; i++) out[i] = lv_color_make(r[i], g[i], b[i]).full;
}

REF_API void ref_color_to32_n(const uint16_t * c, uint32_t * out, size_t n)
{
    for(size_t i = 0; i < n; i++) {
        lv_color_t a;
        a.full = c[i];
        out[i] = lv_color_to32(a);
    }
}

REF_API void ref_color_fill(uint16_t * buf, uint16_t color, uint32_t px_num)
{
    lv_color_t c;
    c.full = color;
    lv_color_fill((lv_color_t *)buf, c, px_num);
}

REF_API void ref_color_mix_with_alpha_n(const uint16_t * bg, const uint8_t * bg_opa, const uint16_t * fg,
                                        const uint8_t * fg_opa, uint16_t * res, uint8_t * res_opa, size_t n)
{
    for(size_t i = 0; i < n; i++) {
        lv_color_t b, f, r;
        lv_opa_t ro;
        b.full = bg[i];
        f.full = fg[i];
        lv_color_mix_with_alpha(b, bg_opa[i], f, fg_opa[i], &r, &ro);
        res[i] = r.full;
        res_opa[i] = ro;
    }
}

/* ------------------------------------------------------------------------------------------------------------ */
/* Binary layout of the boundary types, for the mirror structs of the LVGL-signature shim.                      */
/* ------------------------------------------------------------------------------------------------------------ */

#define LAY(name, expr) \
    if(strcmp(key, name) == 0) return (long)(expr)

REF_API long ref_layout(const char * key)
{
    LAY("sizeof(lv_area_t)", sizeof(lv_area_t));
    LAY("sizeof(lv_color_t)", sizeof(lv_color_t));
    LAY("sizeof(lv_coord_t)", sizeof(lv_coord_t));
    LAY("sizeof(lv_draw_sw_blend_dsc_t)", sizeof(lv_draw_sw_blend_dsc_t));
    LAY("blend_dsc.blend_area", offsetof(lv_draw_sw_blend_dsc_t, blend_area));
    LAY("blend_dsc.src_buf", offsetof(lv_draw_sw_blend_dsc_t, src_buf));
    LAY("blend_dsc.color", offsetof(lv_draw_sw_blend_dsc_t, color));
    LAY("blend_dsc.mask_buf", offsetof(lv_draw_sw_blend_dsc_t, mask_buf));
    LAY("blend_dsc.mask_res", offsetof(lv_draw_sw_blend_dsc_t, mask_res));
    LAY("blend_dsc.mask_area", offsetof(lv_draw_sw_blend_dsc_t, mask_area));
    LAY("blend_dsc.opa", offsetof(lv_draw_sw_blend_dsc_t, opa));
    LAY("blend_dsc.blend_mode", offsetof(lv_draw_sw_blend_dsc_t, blend_mode));
    LAY("sizeof(lv_draw_ctx_t)", sizeof(lv_draw_ctx_t));
    LAY("draw_ctx.buf", offsetof(lv_draw_ctx_t, buf));
    LAY("draw_ctx.buf_area", offsetof(lv_draw_ctx_t, buf_area));
    LAY("draw_ctx.clip_area", offsetof(lv_draw_ctx_t, clip_area));
    LAY("draw_ctx.init_buf", offsetof(lv_draw_ctx_t, init_buf));
    LAY("draw_ctx.wait_for_finish", offsetof(lv_draw_ctx_t, wait_for_finish));
    LAY("draw_ctx.buffer_copy", offsetof(lv_draw_ctx_t, buffer_copy));
    LAY("draw_ctx.layer_instance_size", offsetof(lv_draw_ctx_t, layer_instance_size));
    LAY("draw_ctx.user_data", offsetof(lv_draw_ctx_t, user_data));
    LAY("sizeof(lv_draw_sw_ctx_t)", sizeof(lv_draw_sw_ctx_t));
    LAY("sw_ctx.blend", offsetof(lv_draw_sw_ctx_t, blend));
    LAY("sizeof(lv_disp_draw_buf_t)", sizeof(lv_disp_draw_buf_t));
    LAY("draw_buf.buf1", offsetof(lv_disp_draw_buf_t, buf1));
    LAY("draw_buf.buf2", offsetof(lv_disp_draw_buf_t, buf2));
    LAY("draw_buf.buf_act", offsetof(lv_disp_draw_buf_t, buf_act));
    LAY("draw_buf.size", offsetof(lv_disp_draw_buf_t, size));
    LAY("draw_buf.flushing", offsetof(lv_disp_draw_buf_t, flushing));
    LAY("draw_buf.flushing_last", offsetof(lv_disp_draw_buf_t, flushing_last));
    LAY("sizeof(lv_disp_drv_t)", sizeof(lv_disp_drv_t));
    LAY("disp_drv.hor_res", offsetof(lv_disp_drv_t, hor_res));
    LAY("disp_drv.ver_res", offsetof(lv_disp_drv_t, ver_res));
    LAY("disp_drv.offset_x", offsetof(lv_disp_drv_t, offset_x));
    LAY("disp_drv.offset_y", offsetof(lv_disp_drv_t, offset_y));
    LAY("disp_drv.draw_buf", offsetof(lv_disp_drv_t, draw_buf));
    LAY("disp_drv.flush_cb", offsetof(lv_disp_drv_t, flush_cb));
    LAY("disp_drv.set_px_cb", offsetof(lv_disp_drv_t, set_px_cb));
    LAY("disp_drv.draw_ctx", offsetof(lv_disp_drv_t, draw_ctx));
    LAY("disp_drv.draw_ctx_init", offsetof(lv_disp_drv_t, draw_ctx_init));
    LAY("disp_drv.draw_ctx_deinit", offsetof(lv_disp_drv_t, draw_ctx_deinit));
    LAY("disp_drv.draw_ctx_size", offsetof(lv_disp_drv_t, draw_ctx_size));
    LAY("disp_drv.user_data", offsetof(lv_disp_drv_t, user_data));
    LAY("LV_DRAW_MASK_RES_TRANSP", LV_DRAW_MASK_RES_TRANSP);
    LAY("LV_DRAW_MASK_RES_FULL_COVER", LV_DRAW_MASK_RES_FULL_COVER);
    LAY("LV_DRAW_MASK_RES_CHANGED", LV_DRAW_MASK_RES_CHANGED);
    LAY("LV_BLEND_MODE_NORMAL", LV_BLEND_MODE_NORMAL);
    LAY("LV_BLEND_MODE_ADDITIVE", LV_BLEND_MODE_ADDITIVE);
    LAY("LV_BLEND_MODE_SUBTRACTIVE", LV_BLEND_MODE_SUBTRACTIVE);
    LAY("LV_BLEND_MODE_MULTIPLY", LV_BLEND_MODE_MULTIPLY);
    LAY("LV_DISP_ROT_90", LV_DISP_ROT_90);
    LAY("LV_DISP_ROT_180", LV_DISP_ROT_180);
    LAY("LV_DISP_ROT_270", LV_DISP_ROT_270);
    LAY("LV_IMG_CF_TRUE_COLOR", LV_IMG_CF_TRUE_COLOR);
    LAY("LV_IMG_CF_TRUE_COLOR_ALPHA", LV_IMG_CF_TRUE_COLOR_ALPHA);
    LAY("LV_IMG_CF_ALPHA_8BIT", LV_IMG_CF_ALPHA_8BIT);
    LAY("LV_IMG_CF_RGB565A8", LV_IMG_CF_RGB565A8);
    return -1;
}

/* Bitfield word of lv_disp_drv_t as the compiler lays it out: set the named field to all ones in a zeroed struct
 * and return the 32-bit word at byte offset 24 (the tests compare this with the mirror's masks). */
REF_API uint32_t ref_disp_drv_bits(const char * key)
{
    lv_disp_drv_t d;
    memset(&d, 0, sizeof(d));
    if(strcmp(key, "direct_mode") == 0) d.direct_mode = 1;
    else if(strcmp(key, "full_refresh") == 0) d.full_refresh = 1;
    else if(strcmp(key, "sw_rotate") == 0) d.sw_rotate = 1;
    else if(strcmp(key, "antialiasing") == 0) d.antialiasing = 1;
    else if(strcmp(key, "rotated") == 0) d.rotated = 3;
    else if(strcmp(key, "screen_transp") == 0) d.screen_transp = 1;
    else return 0;
    uint32_t w;
    memcpy(&w, (const uint8_t *)&d + offsetof(lv_disp_drv_t, draw_buf) + sizeof(void *), sizeof(w));
    return w;
}

/* ------------------------------------------------------------------------------------------------------------ */
/* Display 1: a full LVGL display whose draw context / flush callback may be replaced by a plug-in, used for   */
/* the end-to-end "real widgets through the replacement blend" comparison.                                      */
/* ------------------------------------------------------------------------------------------------------------ */

typedef void (*ctx_init_fn)(lv_disp_drv_t *, lv_draw_ctx_t *);
typedef void (*flush_fn)(lv_disp_drv_t *, const lv_area_t *, lv_color_t *);

static lv_disp_draw_buf_t g_dbuf1;
static lv_disp_drv_t g_drv1;
static lv_disp_t * g_disp1;
static lv_color_t * g_buf1;
static lv_color_t * g_buf1b; /* second draw buffer (flags bit 6: lv_disp_draw_buf_init with a buf2), or NULL */
static lv_color_t * g_fb1; /* captured by capture_flush */
static int g_fb_w, g_fb_h;
static int g_flush_calls;
static int g_px_bytes = 2; /* 3 when the scene display has screen_transp (ARGB8565 draw buffer) */

/* What a display driver with its own flush_cb has to call first when the draw context defers its work
 * (the plug-in's wait-for-finish entry point); NULL for LVGL's software context. */
static void (*g_pre_flush)(lv_draw_ctx_t *);

REF_API void ref_scene_set_pre_flush(void * fn)
{
    g_pre_flush = (void (*)(lv_draw_ctx_t *))fn;
}

/* Where the display's draw buffer comes from: the port takes it from heap_caps_malloc with the capabilities the
 * board asks for (DMA-capable memory with flags.buff_dma, src/lv_port.c:198-207) and frees it with free (:256-258,
 * :286-288).  NULL/NULL = calloc/free. */
static void * (*g_buf_alloc)(size_t);
static void (*g_buf_free)(void *);
static void (*g_buf1_free)(void *); /* what frees the current g_buf1 */

REF_API void ref_scene_set_buf_allocator(void * alloc_fn, void * free_fn)
{
    g_buf_alloc = (void * (*)(size_t))alloc_fn;
    g_buf_free = (void (*)(void *))free_fn;
}

static void capture_flush(lv_disp_drv_t * drv, const lv_area_t * a, lv_color_t * p)
{
    if(g_pre_flush) g_pre_flush(drv->draw_ctx);
    int w = a->x2 - a->x1 + 1;
    const uint8_t * src = (const uint8_t *)p;
    uint8_t * fb = (uint8_t *)g_fb1;
    for(int y = a->y1; y <= a->y2; y++) {
        if(y < 0 || y >= g_fb_h) continue;
        memcpy(fb + ((size_t)y * g_fb_w + a->x1) * g_px_bytes, src + (size_t)(y - a->y1) * w * g_px_bytes,
               (size_t)w * g_px_bytes);
    }
    g_flush_calls++;
    lv_disp_flush_ready(drv);
}

/* Register the scene display.  init/deinit/size == NULL/0 keeps LVGL's software context; flush == NULL keeps the
 * capturing flush above.  full_refresh as in the port (src/lv_port.c:242). */
/* flags: bit 0 = anti-aliasing off (masks are thresholded, lv_draw_sw_blend.c:150-156), bit 1 = screen_transp
 * (ARGB8565 draw buffer, 3 bytes per pixel), bit 2 = LVGL's own software rotation (drv.sw_rotate,
 * libraries/lvgl/src/core/lv_refr.c:1084-1265) with drv.rotated = bits 3-4; hor_res/ver_res are then the PHYSICAL
 * size, the scene is laid out on the swapped logical size and the frame buffer collects the rotated chunks */
REF_API int ref_scene_display(int hor_res, int ver_res, void * init, void * deinit, size_t ctx_size, void * flush,
                              void * user_data, int full_refresh, int flags)
{
    if(!g_inited) {
        lv_init();
        g_inited = 1;
    }
    if(g_disp1) {
        lv_disp_remove(g_disp1);
        g_disp1 = NULL;
        if(g_buf1_free) g_buf1_free(g_buf1);
        else free(g_buf1);
        if(g_buf1b) {
            if(g_buf1_free) g_buf1_free(g_buf1b);
            else free(g_buf1b);
            g_buf1b = NULL;
        }
        free(g_fb1);
    }
    g_fb_w = hor_res;
    g_fb_h = ver_res;
    g_px_bytes = (flags & 2) ? LV_IMG_PX_SIZE_ALPHA_BYTE : (int)sizeof(lv_color_t);
    if(g_buf_alloc && g_buf_free) {
        g_buf1 = g_buf_alloc((size_t)hor_res * ver_res * (size_t)g_px_bytes);
        if(!g_buf1) return -1;
        memset(g_buf1, 0, (size_t)hor_res * ver_res * (size_t)g_px_bytes);
        g_buf1_free = g_buf_free;
    }
    else {
        g_buf1 = calloc((size_t)hor_res * ver_res, (size_t)g_px_bytes);
        g_buf1_free = NULL;
    }
    if(flags & 64) { /* two draw buffers that LVGL alternates between (libraries/lvgl/src/hal/lv_hal_disp.c:150-158) */
        const size_t bytes = (size_t)hor_res * ver_res * (size_t)g_px_bytes;
        g_buf1b = g_buf1_free ? g_buf_alloc(bytes) : malloc(bytes);
        if(!g_buf1b) return -1;
        memset(g_buf1b, 0, bytes);
    }
    g_fb1 = calloc((size_t)hor_res * ver_res, (size_t)g_px_bytes);
    g_flush_calls = 0;
    lv_disp_draw_buf_init(&g_dbuf1, g_buf1, g_buf1b, (uint32_t)hor_res * ver_res);
    lv_disp_drv_init(&g_drv1);
    g_drv1.hor_res = (lv_coord_t)hor_res;
    g_drv1.ver_res = (lv_coord_t)ver_res;
    g_drv1.draw_buf = &g_dbuf1;
    g_drv1.full_refresh = full_refresh ? 1 : 0;
    g_drv1.flush_cb = flush ? (flush_fn)flush : capture_flush;
    g_drv1.user_data = user_data;
    if(flags & 1) g_drv1.antialiasing = 0;
    if(flags & 2) g_drv1.screen_transp = 1;
    if(flags & 4) {
        g_drv1.sw_rotate = 1;
        g_drv1.rotated = (flags >> 3) & 3;
        g_drv1.full_refresh = 0; /* lv_refr.c:1178-1181 refuses to rotate a full-refreshed display */
    }
    if(init) {
        g_drv1.draw_ctx_init = (ctx_init_fn)init;
        g_drv1.draw_ctx_deinit = (ctx_init_fn)deinit;
        g_drv1.draw_ctx_size = ctx_size;
    }
    g_disp1 = lv_disp_drv_register(&g_drv1);
    if(!g_disp1) return -1;
    lv_disp_set_default(g_disp1);
    return 0;
}

REF_API void * ref_scene_fb(void)
{
    return g_fb1;
}
REF_API void * ref_scene_draw_buf(void)
{
    return g_buf1;
}
REF_API void * ref_scene_draw_buf_act(void) /* the buffer LVGL drew the last refresh into */
{
    return g_dbuf1.buf_act;
}
REF_API int ref_scene_px_bytes(void)
{
    return g_px_bytes;
}
REF_API int ref_scene_flush_calls(void)
{
    return g_flush_calls;
}
REF_API void * ref_scene_drv(void)
{
    return &g_drv1;
}

/* Deterministic procedural image with alpha (LV_IMG_CF_TRUE_COLOR_ALPHA, 3 bytes/px at depth 16). */
static uint8_t * g_img_data;
static lv_img_dsc_t g_img;

static void make_img(int w, int h)
{
    free(g_img_data);
    g_img_data = malloc((size_t)w * h * 3);
    uint32_t s = 0x1234567u;
    for(int y = 0; y < h; y++) {
        for(int x = 0; x < w; x++) {
            s = s * 1664525u + 1013904223u;
            lv_color_t c = lv_color_make((uint8_t)(x * 255 / w), (uint8_t)(y * 255 / h), (uint8_t)(s >> 24));
            int dx = x - w / 2, dy = y - h / 2;
            int d = dx * dx + dy * dy;
            int r2 = (w / 2) * (w / 2);
            int a = d >= r2 ? 0 : 255 - (d * 255 / r2);
            if(((x / 8) + (y / 8)) % 5 == 0) a = 255;
            uint8_t * p = g_img_data + ((size_t)y * w + x) * 3;
            p[0] = (uint8_t)(c.full & 0xFF);
            p[1] = (uint8_t)(c.full >> 8);
            p[2] = (uint8_t)a;
        }
    }
    memset(&g_img, 0, sizeof(g_img));
    g_img.header.always_zero = 0;
    g_img.header.cf = LV_IMG_CF_TRUE_COLOR_ALPHA;
    g_img.header.w = (uint32_t)w;
    g_img.header.h = (uint32_t)h;
    g_img.data_size = (uint32_t)(w * h * 3);
    g_img.data = g_img_data;
}

/* Build a widget scene on the scene display's active screen.  Everything here is stock LVGL API usage. */
REF_API int ref_scene_build(int id)
{
    if(!g_disp1) return -1;
    lv_disp_set_default(g_disp1);
    lv_obj_t * scr = lv_disp_get_scr_act(g_disp1);
    lv_obj_clean(scr);
    lv_obj_set_style_bg_color(scr, lv_color_hex(0x203040), 0);
    lv_obj_set_style_bg_opa(scr, g_drv1.screen_transp ? LV_OPA_60 : LV_OPA_COVER, 0);
    if(g_drv1.screen_transp) lv_disp_set_bg_opa(g_disp1, LV_OPA_TRANSP);
    int W = lv_disp_get_hor_res(g_disp1), H = lv_disp_get_ver_res(g_disp1); /* logical size */

    if(id == 0 || id == 2) {
        /* opaque + translucent rounded rectangles, gradient, border, shadow */
        lv_obj_t * a = lv_obj_create(scr);
        lv_obj_set_size(a, W / 2, H / 3);
        lv_obj_set_pos(a, 10, 10);
        lv_obj_set_style_radius(a, 12, 0);
        lv_obj_set_style_bg_color(a, lv_color_hex(0xE04020), 0);
        lv_obj_set_style_bg_opa(a, LV_OPA_70, 0);
        lv_obj_set_style_border_width(a, 3, 0);
        lv_obj_set_style_border_color(a, lv_color_hex(0xFFFFFF), 0);
        lv_obj_set_style_border_opa(a, LV_OPA_50, 0);

        lv_obj_t * b = lv_obj_create(scr);
        lv_obj_set_size(b, W / 2, H / 3);
        lv_obj_set_pos(b, W / 3, H / 4);
        lv_obj_set_style_radius(b, LV_RADIUS_CIRCLE, 0);
        lv_obj_set_style_bg_color(b, lv_color_hex(0x20C060), 0);
        lv_obj_set_style_bg_grad_color(b, lv_color_hex(0x2040F0), 0);
        lv_obj_set_style_bg_grad_dir(b, LV_GRAD_DIR_VER, 0);
        lv_obj_set_style_bg_opa(b, LV_OPA_COVER, 0);
        lv_obj_set_style_shadow_width(b, 15, 0);
        lv_obj_set_style_shadow_color(b, lv_color_hex(0x000000), 0);
        lv_obj_set_style_shadow_opa(b, LV_OPA_60, 0);

        lv_obj_t * c = lv_obj_create(scr);
        lv_obj_set_size(c, W - 40, 30);
        lv_obj_set_pos(c, 20, H - 50);
        lv_obj_set_style_radius(c, 0, 0);
        lv_obj_set_style_bg_color(c, lv_color_hex(0xF0F000), 0);
        lv_obj_set_style_bg_opa(c, LV_OPA_30, 0);
        lv_obj_set_style_border_width(c, 0, 0);
    }
    if(id == 1 || id == 2) {
        /* text, arc, bar, button, line, alpha image */
        lv_obj_t * l = lv_label_create(scr);
        lv_label_set_text(l, "Damper 42% 187.5 C");
        lv_obj_set_style_text_color(l, lv_color_hex(0xFFE080), 0);
        lv_obj_set_pos(l, 12, H / 2);

        lv_obj_t * arc = lv_arc_create(scr);
        lv_obj_set_size(arc, 100, 100);
        lv_obj_set_pos(arc, W - 120, 10);
        lv_arc_set_value(arc, 63);

        lv_obj_t * bar = lv_bar_create(scr);
        lv_obj_set_size(bar, W / 2, 14);
        lv_obj_set_pos(bar, 14, H / 2 + 30);
        lv_bar_set_value(bar, 70, LV_ANIM_OFF);

        lv_obj_t * btn = lv_btn_create(scr);
        lv_obj_set_pos(btn, 14, H / 2 + 60);
        lv_obj_t * bl = lv_label_create(btn);
        lv_label_set_text(bl, "AUTO");

        static lv_point_t pts[3];
        pts[0].x = 5;
        pts[0].y = 5;
        pts[1].x = (lv_coord_t)(W - 30);
        pts[1].y = (lv_coord_t)(H / 3);
        pts[2].x = 40;
        pts[2].y = (lv_coord_t)(H - 20);
        lv_obj_t * line = lv_line_create(scr);
        lv_line_set_points(line, pts, 3);
        lv_obj_set_style_line_width(line, 5, 0);
        lv_obj_set_style_line_color(line, lv_color_hex(0xFF40FF), 0);
        lv_obj_set_style_line_opa(line, LV_OPA_80, 0);
        lv_obj_set_style_line_rounded(line, 1, 0);

        make_img(96, 80);
        lv_obj_t * img = lv_img_create(scr);
        lv_img_set_src(img, &g_img);
        lv_obj_set_pos(img, W / 2 - 20, H / 2 + 20);
        lv_obj_t * img2 = lv_img_create(scr);
        lv_img_set_src(img2, &g_img);
        lv_obj_set_pos(img2, W / 2 + 40, H / 2 - 40);
        lv_obj_set_style_img_opa(img2, LV_OPA_60, 0);
    }
    if(id == 3) {
        /* layers: a half-transparent container with children (simple layer, lv_refr.c layer_type SIMPLE), a rotated and
         * zoomed object (transform layer with alpha) and a non-normal blend mode on a widget */
        lv_obj_t * box = lv_obj_create(scr);
        lv_obj_set_size(box, W / 2, H / 2);
        lv_obj_set_pos(box, 8, 8);
        lv_obj_set_style_bg_color(box, lv_color_hex(0x3060C0), 0);
        lv_obj_set_style_radius(box, 10, 0);
        lv_obj_set_style_opa(box, LV_OPA_50, 0);
        lv_obj_t * inner = lv_label_create(box);
        lv_label_set_text(inner, "layered 50%");
        lv_obj_t * ib = lv_btn_create(box);
        lv_obj_set_pos(ib, 10, 30);
        lv_obj_t * ibl = lv_label_create(ib);
        lv_label_set_text(ibl, "OK");

        lv_obj_t * rot = lv_obj_create(scr);
        lv_obj_set_size(rot, 90, 50);
        lv_obj_set_pos(rot, W / 2, H / 3);
        lv_obj_set_style_bg_color(rot, lv_color_hex(0xE0A020), 0);
        lv_obj_set_style_transform_angle(rot, 300, 0);
        lv_obj_set_style_transform_zoom(rot, 300, 0);
        lv_obj_t * rl = lv_label_create(rot);
        lv_label_set_text(rl, "rot");

        lv_obj_t * add = lv_obj_create(scr);
        lv_obj_set_size(add, W / 3, H / 4);
        lv_obj_set_pos(add, W / 4, H / 2);
        lv_obj_set_style_bg_color(add, lv_color_hex(0x404040), 0);
        lv_obj_set_style_blend_mode(add, LV_BLEND_MODE_ADDITIVE, 0);
    }
    lv_obj_invalidate(scr);
    return 0;
}

REF_API void ref_scene_refresh(void)
{
    lv_refr_now(g_disp1);
}

/* ------------------------------------------------------------------------------------------------------------ */
/* Recorder: a draw context that is LVGL's software context with ->blend wrapped, so that every descriptor the    */
/* producers issue for a scene (libraries/lvgl/src/draw/sw/lv_draw_sw_blend.c:95-106 hands them to ctx->blend) is  */
/* kept, already clipped the way lv_draw_sw_blend_basic clips it (:108-166), with private copies of the source    */
/* and mask rectangles.  The drawing itself still happens in LVGL's own blend.                                    */
/* ------------------------------------------------------------------------------------------------------------ */

typedef struct {
    int16_t area[4];  /* blend_area intersected with the clip area */
    uint16_t * src;   /* w*h pixels or NULL (fill) */
    uint8_t * mask;   /* w*h bytes or NULL */
    uint16_t color;
    uint8_t opa;
    uint8_t blend_mode;
} ref_rec_t;

static ref_rec_t * g_rec;
static int g_rec_n, g_rec_cap;
static void (*g_rec_orig)(lv_draw_ctx_t *, const lv_draw_sw_blend_dsc_t *);

static void rec_blend(lv_draw_ctx_t * ctx, const lv_draw_sw_blend_dsc_t * dsc)
{
    g_rec_orig(ctx, dsc); /* first: with anti-aliasing off it thresholds the mask in place, and that is what counts */
    const lv_opa_t * mask = dsc->mask_buf;
    if(mask && dsc->mask_res == LV_DRAW_MASK_RES_TRANSP) return;
    if(dsc->mask_res == LV_DRAW_MASK_RES_FULL_COVER) mask = NULL;
    lv_area_t a;
    if(!_lv_area_intersect(&a, dsc->blend_area, ctx->clip_area)) return;
    int w = lv_area_get_width(&a), h = lv_area_get_height(&a);
    if(g_rec_n == g_rec_cap) {
        g_rec_cap = g_rec_cap ? g_rec_cap * 2 : 256;
        g_rec = realloc(g_rec, (size_t)g_rec_cap * sizeof(ref_rec_t));
    }
    ref_rec_t * r = &g_rec[g_rec_n++];
    memset(r, 0, sizeof(*r));
    r->area[0] = a.x1;
    r->area[1] = a.y1;
    r->area[2] = a.x2;
    r->area[3] = a.y2;
    r->color = dsc->color.full;
    r->opa = dsc->opa;
    r->blend_mode = (uint8_t)dsc->blend_mode;
    if(dsc->src_buf) {
        int stride = lv_area_get_width(dsc->blend_area);
        const lv_color_t * s = dsc->src_buf + (size_t)stride * (a.y1 - dsc->blend_area->y1) + (a.x1 - dsc->blend_area->x1);
        r->src = malloc((size_t)w * h * 2);
        for(int y = 0; y < h; y++) memcpy(r->src + (size_t)y * w, s + (size_t)y * stride, (size_t)w * 2);
    }
    if(mask) {
        int stride = lv_area_get_width(dsc->mask_area);
        const lv_opa_t * m = mask + (size_t)stride * (a.y1 - dsc->mask_area->y1) + (a.x1 - dsc->mask_area->x1);
        r->mask = malloc((size_t)w * h);
        for(int y = 0; y < h; y++) memcpy(r->mask + (size_t)y * w, m + (size_t)y * stride, (size_t)w);
    }
}

REF_API void ref_record_ctx_init(lv_disp_drv_t * drv, lv_draw_ctx_t * ctx)
{
    lv_draw_sw_init_ctx(drv, ctx);
    lv_draw_sw_ctx_t * sw = (lv_draw_sw_ctx_t *)ctx;
    g_rec_orig = sw->blend;
    sw->blend = rec_blend;
}
REF_API void ref_record_ctx_deinit(lv_disp_drv_t * drv, lv_draw_ctx_t * ctx)
{
    lv_draw_sw_deinit_ctx(drv, ctx);
}
REF_API size_t ref_record_ctx_size(void)
{
    return sizeof(lv_draw_sw_ctx_t);
}
REF_API void ref_record_reset(void)
{
    for(int i = 0; i < g_rec_n; i++) {
        free(g_rec[i].src);
        free(g_rec[i].mask);
    }
    g_rec_n = 0;
}
REF_API int ref_record_count(void)
{
    return g_rec_n;
}
REF_API const ref_rec_t * ref_record_items(void)
{
    return g_rec;
}
