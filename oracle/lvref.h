/*
 * lvref.h -- CPU oracle (TEST INFRASTRUCTURE ONLY; never linked, loaded or called by the product path).
 *
 * A from-scratch restatement, in plain C, of the arithmetic of LVGL 8.x's software pixel path as configured by the
 * reference firmware (depth 16, LV_COLOR_MIX_ROUND_OFS 0, LV_COLOR_16_SWAP selectable at run time).  Each function
 * cites the reference lines it follows.  Pinning: tests/test_oracle_vs_reference.py checks every function here
 * against the real LVGL compiled from the reference sources (oracle/_ref), and tests/golden/ holds outputs of that
 * real LVGL for the committed cases.  The flush restatement (lvref_flush*) has no such pin -- src/lv_port.c needs
 * ESP-IDF headers that do not exist in the build image -- and is "parity unpinned".
 */
#ifndef LVREF_H
#define LVREF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int16_t x1, y1, x2, y2;
} lvref_area_t;

/* ---- colour arithmetic (libraries/lvgl/src/misc/lv_color.h) ---- */
uint16_t lvref_color_mix(uint16_t c1, uint16_t c2, uint8_t mix, int swap);                 /* :438-472 */
void lvref_color_premult(uint16_t c, uint8_t mix, uint16_t out[3], int swap);              /* :474-488 */
uint16_t lvref_color_mix_premult(const uint16_t pm[3], uint16_t c2, uint8_t mix, int swap); /* :499-519 */
uint16_t lvref_color_make(uint8_t r, uint8_t g, uint8_t b, int swap);                      /* :148-154, :594-597 */
uint16_t lvref_color_hex(uint32_t c, int swap);                                            /* :599-631 */
uint32_t lvref_color_to32(uint16_t c, int swap);                                           /* :374-427 */
void lvref_color_fill(uint16_t * buf, uint16_t color, uint32_t px_num);                    /* lv_color.c:36-77 */
void lvref_color_mix_with_alpha(uint16_t bg, uint8_t bg_opa, uint16_t fg, uint8_t fg_opa, uint16_t * res,
                                uint8_t * res_opa, int swap);                              /* :530-577 */

void lvref_color_mix_n(const uint16_t * c1, const uint16_t * c2, const uint8_t * mix, uint16_t * out, size_t n,
                       int swap);
uint64_t lvref_color_mix_digest(int mix, uint32_t fg_lo, uint32_t fg_hi, int swap);
void lvref_color_hex_n(const uint32_t * c, uint16_t * out, size_t n, int swap);
void lvref_color_to32_n(const uint16_t * c, uint32_t * out, size_t n, int swap);

/* ---- area (libraries/lvgl/src/misc/lv_area.c:129-144) ---- */
int lvref_area_intersect(lvref_area_t * res, const lvref_area_t * a, const lvref_area_t * b);

/* ---- blend (libraries/lvgl/src/draw/sw/lv_draw_sw_blend.c) ---- */
typedef struct {
    void * buf;                  /* draw_ctx->buf */
    lvref_area_t buf_area;       /* draw_ctx->buf_area */
    lvref_area_t clip_area;      /* draw_ctx->clip_area */
    lvref_area_t blend_area;     /* dsc->blend_area */
    const uint16_t * src_buf;    /* dsc->src_buf, NULL = fill */
    uint16_t color;              /* dsc->color.full */
    const uint8_t * mask_buf;    /* dsc->mask_buf or NULL */
    uint8_t mask_res;            /* dsc->mask_res */
    lvref_area_t mask_area;      /* dsc->mask_area */
    uint8_t opa;                 /* dsc->opa */
    uint8_t blend_mode;          /* dsc->blend_mode */
    uint8_t swap;                /* LV_COLOR_16_SWAP */
    uint8_t screen_transp;       /* disp->driver->screen_transp (ARGB8565 destination) */
    uint8_t antialiasing_off;    /* disp->driver->antialiasing == 0: the mask is rounded to 0 / 255, IN PLACE over the
                                    whole mask_area, before anything else looks at it (:150-156) */
} lvref_blend_t;

/* lv_draw_sw_blend (:95-106) + lv_draw_sw_blend_basic (:108-204).  Returns 0, or -1 for a mode not restated. */
int lvref_blend(const lvref_blend_t * b);

/* ---- image colour/alpha split ---- */
/* format: 2 = ARGB8888 bytes {B,G,R,A}, 3 = RGBA8888 bytes {R,G,B,A}, 4 = ARGB8565 3 B/px {c.lo,c.hi,A}
 * (libraries/lvgl/src/extra/libs/png/lv_png.c:244-253, libraries/lvgl/src/draw/sw/lv_draw_sw_img.c:249-275). */
void lvref_image_split(const void * src, int format, uint16_t * color, uint8_t * alpha, size_t n, int swap);
void lvref_image_convert(const void * src, int format, uint16_t * color, uint8_t * alpha, size_t n, int swap,
                         int chroma_keyed, uint16_t chroma_key, uint16_t recolor, uint8_t recolor_opa);

/* An image with per-pixel alpha drawn over a buffer the way lv_draw_sw_img_decoded's generic branch does it
 * (libraries/lvgl/src/draw/sw/lv_draw_sw_img.c:98-195): split into colour + alpha for at most hor_res pixels'
 * worth of whole rows at a time, one lv_draw_sw_blend per chunk.  coords = image position; src covers coords. */
int lvref_img_blend(void * buf, lvref_area_t buf_area, lvref_area_t clip_area, lvref_area_t coords, const void * src,
                    int format, uint8_t opa, int swap, int hor_res);
/* The same on a display with anti-aliasing switched off: every chunk's alpha plane is the mask of its blend and is
 * rounded there (libraries/lvgl/src/draw/sw/lv_draw_sw_blend.c:150-156). */
int lvref_img_blend_aa(void * buf, lvref_area_t buf_area, lvref_area_t clip_area, lvref_area_t coords, const void * src,
                       int format, uint8_t opa, int swap, int hor_res, int antialiasing_off);

/* ---- flush (src/lv_port.c:428-572) ---- */
typedef void (*lvref_draw_bitmap_cb)(void * user, int x_start, int y_start, int x_end, int y_end,
                                     const uint16_t * color_data);
/* The callback's loop nest with its transport chunking; cb stands in for esp_lcd_panel_draw_bitmap.
 * hor_res/ver_res are drv->hor_res/ver_res.  Returns the number of chunks sent. */
int lvref_flush(const lvref_area_t * area, const uint16_t * color_map, int rotation, uint32_t trans_size, int hor_res,
                int ver_res, lvref_draw_bitmap_cb cb, void * user);
/* Same pixels as the concatenation of lvref_flush's chunks for one area, written to out (width*height px). */
void lvref_flush_frame(const uint16_t * in, uint16_t * out, int width, int height, int rotation, int swap_bytes);

/* ---- panel driver (src/esp_lcd_axs15231b.c:173-193, 292-337): what one draw_bitmap hands to the panel IO layer.
 * tx stands in for esp_lcd_panel_io_tx_param (is_color 0) / esp_lcd_panel_io_tx_color (is_color 1); size in bytes.
 * Parity unpinned: the driver needs ESP-IDF headers that are not in the reference tree, so it cannot be compiled
 * here; this is a line-by-line restatement. ---- */
typedef void (*lvref_panel_tx_cb)(void * user, int is_color, int lcd_cmd, const uint8_t * param, size_t size);
void lvref_axs15231b_draw_bitmap(int x_start, int y_start, int x_end, int y_end, int x_gap, int y_gap, int qspi,
                                 int fb_bits_per_pixel, lvref_panel_tx_cb tx, void * user);

/* ---- synthetic inputs / checksum (shared definition with the GPU library's generator) ---- */
void lvref_synth(void * out, int kind, uint64_t seed, uint64_t first_index, size_t count);
uint64_t lvref_checksum(const void * in, int elem_bytes, uint64_t first_index, size_t count);

/* ---- batch drivers for the CPU baseline (OpenMP over frames) ---- */
/* ARGB8888 image over RGB565 frames (opa 255), then optional flush rotate into out (NULL = blend only). */
void lvref_argb_blend_flush_batch(uint16_t * dst, const uint32_t * src, uint16_t * out, int width, int height,
                                  int n_frames, int rotation, int swap, int threads);

#ifdef __cplusplus
}
#endif
void lvref_list_flush_batch(uint16_t * dst, const lvref_blend_t * items, int n_items, uint16_t * out, int width, int height,
                            int n_frames, int rotation, int threads);

#endif
