/*
 * lvref.c -- CPU oracle (TEST INFRASTRUCTURE ONLY; never linked, loaded or called by the product path).
 * See lvref.h for scope and pinning.  All citations are relative to /root/reference.
 */
#include "lvref.h"

#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define OPA_MIN 2   /* LV_OPA_MIN, libraries/lvgl/src/misc/lv_color.h:57 */
#define OPA_MAX 253 /* LV_OPA_MAX, libraries/lvgl/src/misc/lv_color.h:58 */
#define OPA_COVER 255

static inline uint16_t bswap16(uint16_t v)
{
    return (uint16_t)((v << 8) | (v >> 8));
}

/* Channel access.  With LV_COLOR_16_SWAP the bitfields are green_h:3, red:5, blue:5, green_l:3 from the LSB
 * (libraries/lvgl/src/misc/lv_color.h:208-222), which is plain RGB565 after exchanging the two bytes. */
static inline uint16_t to_native(uint16_t full, int swap)
{
    return swap ? bswap16(full) : full;
}
static inline unsigned ch_r(uint16_t full, int swap)
{
    return to_native(full, swap) >> 11;
}
static inline unsigned ch_g(uint16_t full, int swap)
{
    return (to_native(full, swap) >> 5) & 0x3F;
}
static inline unsigned ch_b(uint16_t full, int swap)
{
    return to_native(full, swap) & 0x1F;
}
/* LV_COLOR_SET_R16/G16/B16 mask to the field width (libraries/lvgl/src/misc/lv_color.h:130-137). */
static inline uint16_t from_rgb(unsigned r, unsigned g, unsigned b, int swap)
{
    uint16_t v = (uint16_t)(((r & 0x1F) << 11) | ((g & 0x3F) << 5) | (b & 0x1F));
    return swap ? bswap16(v) : v;
}

/* libraries/lvgl/src/misc/lv_color.h:438-456 (depth 16, LV_COLOR_MIX_ROUND_OFS 0 branch). */
uint16_t lvref_color_mix(uint16_t c1, uint16_t c2, uint8_t mix, int swap)
{
    if(swap) {
        c1 = bswap16(c1);
        c2 = bswap16(c2);
    }
    uint32_t m = ((uint32_t)mix + 4) >> 3;
    uint32_t bg = ((uint32_t)c2 | ((uint32_t)c2 << 16)) & 0x07E0F81Fu;
    uint32_t fg = ((uint32_t)c1 | ((uint32_t)c1 << 16)) & 0x07E0F81Fu;
    uint32_t res = ((((fg - bg) * m) >> 5) + bg) & 0x07E0F81Fu;
    uint16_t ret = (uint16_t)((res >> 16) | res);
    return swap ? bswap16(ret) : ret;
}

/* libraries/lvgl/src/misc/lv_color.h:474-488 */
void lvref_color_premult(uint16_t c, uint8_t mix, uint16_t out[3], int swap)
{
    out[0] = (uint16_t)(ch_r(c, swap) * mix);
    out[1] = (uint16_t)(ch_g(c, swap) * mix);
    out[2] = (uint16_t)(ch_b(c, swap) * mix);
}

/* LV_UDIV255, libraries/lvgl/src/misc/lv_math.h:132 */
static inline uint32_t udiv255(uint32_t x)
{
    return (x * 0x8081u) >> 0x17;
}

/* libraries/lvgl/src/misc/lv_color.h:499-519 */
uint16_t lvref_color_mix_premult(const uint16_t pm[3], uint16_t c2, uint8_t mix, int swap)
{
    unsigned r = udiv255((uint32_t)pm[0] + ch_r(c2, swap) * mix);
    unsigned g = udiv255((uint32_t)pm[1] + ch_g(c2, swap) * mix);
    unsigned b = udiv255((uint32_t)pm[2] + ch_b(c2, swap) * mix);
    return from_rgb(r, g, b, swap);
}

/* LV_COLOR_MAKE16, libraries/lvgl/src/misc/lv_color.h:148-154: truncation to 5/6/5 bits. */
uint16_t lvref_color_make(uint8_t r, uint8_t g, uint8_t b, int swap)
{
    return from_rgb(r >> 3, g >> 2, b >> 3, swap);
}

/* libraries/lvgl/src/misc/lv_color.h:599-631 */
uint16_t lvref_color_hex(uint32_t c, int swap)
{
    if(!swap) return (uint16_t)(((c & 0xF80000) >> 8) | ((c & 0xFC00) >> 5) | ((c & 0xFF) >> 3));
    return (uint16_t)(((c & 0xF80000) >> 16) | ((c & 0xFC00) >> 13) | ((c & 0x1C00) << 3) | ((c & 0xF8) << 5));
}

/* libraries/lvgl/src/misc/lv_color.h:418-423 */
uint32_t lvref_color_to32(uint16_t c, int swap)
{
    uint32_t r = (ch_r(c, swap) * 263 + 7) >> 5;
    uint32_t g = (ch_g(c, swap) * 259 + 3) >> 6;
    uint32_t b = (ch_b(c, swap) * 263 + 7) >> 5;
    return (b & 0xFF) | ((g & 0xFF) << 8) | ((r & 0xFF) << 16) | 0xFF000000u;
}

/* libraries/lvgl/src/misc/lv_color.c:36-77: the result is px_num copies of color. */
void lvref_color_fill(uint16_t * buf, uint16_t color, uint32_t px_num)
{
    for(uint32_t i = 0; i < px_num; i++) buf[i] = color;
}

/* libraries/lvgl/src/misc/lv_color.h:530-577 (the statics there only memoise) */
void lvref_color_mix_with_alpha(uint16_t bg, uint8_t bg_opa, uint16_t fg, uint8_t fg_opa, uint16_t * res,
                                uint8_t * res_opa, int swap)
{
    if(fg_opa >= OPA_MAX || bg_opa <= OPA_MIN) {
        *res = fg;
        *res_opa = fg_opa;
    }
    else if(fg_opa <= OPA_MIN) {
        *res = bg;
        *res_opa = bg_opa;
    }
    else if(bg_opa >= OPA_MAX) {
        *res = lvref_color_mix(fg, bg, fg_opa, swap);
        *res_opa = OPA_COVER;
    }
    else {
        uint8_t ro = (uint8_t)(255 - ((uint16_t)((uint16_t)(255 - fg_opa) * (255 - bg_opa)) >> 8));
        uint8_t ratio = (uint8_t)((uint16_t)((uint16_t)fg_opa * 255) / ro);
        *res = lvref_color_mix(fg, bg, ratio, swap);
        *res_opa = ro;
    }
}

void lvref_color_mix_n(const uint16_t * c1, const uint16_t * c2, const uint8_t * mix, uint16_t * out, size_t n,
                       int swap)
{
    for(size_t i = 0; i < n; i++) out[i] = lvref_color_mix(c1[i], c2[i], mix[i], swap);
}

uint64_t lvref_color_mix_digest(int mix, uint32_t fg_lo, uint32_t fg_hi, int swap)
{
    uint64_t h = 1469598103934665603ULL;
    for(uint32_t fg = fg_lo; fg < fg_hi; fg++) {
        for(uint32_t bg = 0; bg < 65536; bg++) {
            uint16_t r = lvref_color_mix((uint16_t)fg, (uint16_t)bg, (uint8_t)mix, swap);
            h = (h ^ (r & 0xFF)) * 1099511628211ULL;
            h = (h ^ (r >> 8)) * 1099511628211ULL;
        }
    }
    return h;
}

void lvref_color_hex_n(const uint32_t * c, uint16_t * out, size_t n, int swap)
{
    for(size_t i = 0; i < n; i++) out[i] = lvref_color_hex(c[i], swap);
}

void lvref_color_to32_n(const uint16_t * c, uint32_t * out, size_t n, int swap)
{
    for(size_t i = 0; i < n; i++) out[i] = lvref_color_to32(c[i], swap);
}

/* libraries/lvgl/src/misc/lv_area.c:129-144 */
int lvref_area_intersect(lvref_area_t * res, const lvref_area_t * a, const lvref_area_t * b)
{
    res->x1 = a->x1 > b->x1 ? a->x1 : b->x1;
    res->y1 = a->y1 > b->y1 ? a->y1 : b->y1;
    res->x2 = a->x2 < b->x2 ? a->x2 : b->x2;
    res->y2 = a->y2 < b->y2 ? a->y2 : b->y2;
    return !(res->x1 > res->x2 || res->y1 > res->y2);
}

static inline int aw(const lvref_area_t * a)
{
    return a->x2 - a->x1 + 1;
}
static inline int ah(const lvref_area_t * a)
{
    return a->y2 - a->y1 + 1;
}

/* ------------------------------------------------------------------------------------------------------------ */
/* fill_normal, libraries/lvgl/src/draw/sw/lv_draw_sw_blend.c:242-375                                           */
/* ------------------------------------------------------------------------------------------------------------ */
static void fill_normal(uint16_t * dest, int w, int h, int dest_stride, uint16_t color, uint8_t opa,
                        const uint8_t * mask, int mask_stride, int swap)
{
    if(mask == NULL) {
        if(opa >= OPA_MAX) { /* :254-259 */
            for(int y = 0; y < h; y++) lvref_color_fill(dest + (size_t)y * dest_stride, color, (uint32_t)w);
        }
        else { /* :261-287: one-entry cache that starts out holding (black -> lv_color_mix with the ORIGINAL opa) */
            uint16_t last_dest = 0; /* lv_color_black().full == 0 in both byte orders */
            uint16_t last_res = lvref_color_mix(color, last_dest, opa, swap);
            uint8_t opa_q = (uint8_t)((((uint32_t)opa + 4) >> 3) << 3); /* :269-270, wraps to 0 for opa 252 */
            uint16_t pm[3];
            lvref_color_premult(color, opa_q, pm, swap);
            uint8_t opa_inv = (uint8_t)(255 - opa_q);
            for(int y = 0; y < h; y++) {
                uint16_t * row = dest + (size_t)y * dest_stride;
                for(int x = 0; x < w; x++) {
                    if(last_dest != row[x]) {
                        last_dest = row[x];
                        last_res = lvref_color_mix_premult(pm, row[x], opa_inv, swap);
                    }
                    row[x] = last_res;
                }
            }
        }
    }
    else if(opa >= OPA_MAX) { /* :295-344; the 4-pixel shortcuts there give the same pixels */
        for(int y = 0; y < h; y++) {
            uint16_t * row = dest + (size_t)y * dest_stride;
            const uint8_t * m = mask + (size_t)y * mask_stride;
            for(int x = 0; x < w; x++) {
                if(m[x] == OPA_COVER) row[x] = color;
                else row[x] = lvref_color_mix(color, row[x], m[x], swap);
            }
        }
    }
    else { /* :346-373; last_* there only memoise */
        for(int y = 0; y < h; y++) {
            uint16_t * row = dest + (size_t)y * dest_stride;
            const uint8_t * m = mask + (size_t)y * mask_stride;
            for(int x = 0; x < w; x++) {
                if(m[x] == 0) continue;
                uint8_t o = m[x] == OPA_COVER ? opa : (uint8_t)(((uint32_t)m[x] * opa) >> 8);
                if(o == OPA_COVER) row[x] = color;
                else row[x] = lvref_color_mix(color, row[x], o, swap);
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------------------------ */
/* map_normal, libraries/lvgl/src/draw/sw/lv_draw_sw_blend.c:641-734                                            */
/* ------------------------------------------------------------------------------------------------------------ */
static void map_normal(uint16_t * dest, int w, int h, int dest_stride, const uint16_t * src, int src_stride,
                       uint8_t opa, const uint8_t * mask, int mask_stride, int swap)
{
    for(int y = 0; y < h; y++) {
        uint16_t * d = dest + (size_t)y * dest_stride;
        const uint16_t * s = src + (size_t)y * src_stride;
        const uint8_t * m = mask ? mask + (size_t)y * mask_stride : NULL;
        if(m == NULL) {
            if(opa >= OPA_MAX) memcpy(d, s, (size_t)w * 2); /* :655-660 */
            else
                for(int x = 0; x < w; x++) d[x] = lvref_color_mix(s[x], d[x], opa, swap); /* :662-670 */
        }
        else if(opa > OPA_MAX) { /* :675-718, note the strict '>' */
            for(int x = 0; x < w; x++) {
                if(m[x] == 0) continue;
                if(m[x] == OPA_COVER) d[x] = s[x];
                else d[x] = lvref_color_mix(s[x], d[x], m[x], swap);
            }
        }
        else { /* :720-732 */
            for(int x = 0; x < w; x++) {
                if(m[x] == 0) continue;
                uint8_t o = m[x] >= OPA_MAX ? opa : (uint8_t)(((uint32_t)opa * m[x]) >> 8);
                d[x] = lvref_color_mix(s[x], d[x], o, swap);
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------------------------ */
/* Non-normal blend modes, libraries/lvgl/src/draw/sw/lv_draw_sw_blend.c:950-1050                               */
/* ------------------------------------------------------------------------------------------------------------ */
static uint16_t blend_mode_px(int mode, uint16_t fg, uint16_t bg, uint8_t opa, int swap)
{
    if(opa <= OPA_MIN) return bg;
    int fr = (int)ch_r(fg, swap), fgn = (int)ch_g(fg, swap), fb = (int)ch_b(fg, swap);
    int br = (int)ch_r(bg, swap), bgn = (int)ch_g(bg, swap), bb = (int)ch_b(bg, swap);
    unsigned r, g, b;
    if(mode == 1) { /* additive :950-1001 */
        r = (unsigned)(br + fr > 31 ? 31 : br + fr);
        g = (unsigned)(bgn + fgn > 63 ? 63 : bgn + fgn);
        b = (unsigned)(bb + fb > 31 ? 31 : bb + fb);
    }
    else if(mode == 2) { /* subtractive :1003-1027 */
        r = (unsigned)(br - fr < 0 ? 0 : br - fr);
        b = (unsigned)(bb - fb < 0 ? 0 : bb - fb);
        if(!swap) g = (unsigned)(bgn - fgn < 0 ? 0 : bgn - fgn); /* :1012-1013 */
        else g = (unsigned)(bgn + fgn);                          /* :1015-1018: the swapped build ADDS, then keeps 6 bits */
    }
    else { /* multiply :1029-1050 */
        r = (unsigned)((fr * br) >> 5);
        b = (unsigned)((fb * bb) >> 5);
        g = (unsigned)((fgn * bgn) >> 6);
        if(swap) {
            /* :1040 goes through LV_COLOR_SET_G16 of the swapped build (lv_color.h:134), a macro that evaluates its
             * value argument twice: green_h is stored first, and the second evaluation, which yields green_l, then
             * reads the foreground's green with the NEW high bits and the old low bits. */
            unsigned gh = (g >> 3) & 7u;
            unsigned g2 = (((gh << 3) + ((unsigned)fgn & 7u)) * (unsigned)bgn) >> 6;
            g = (gh << 3) | (g2 & 7u);
        }
    }
    uint16_t c = from_rgb(r, g, b, swap);
    if(opa == OPA_COVER) return c;
    return lvref_color_mix(c, bg, opa, swap);
}

/* fill_blended :534-602 and map_blended :878-948; the last_* variables there only memoise. */
static int fill_or_map_blended(uint16_t * dest, int w, int h, int dest_stride, const uint16_t * src, int src_stride,
                               uint16_t color, uint8_t opa, const uint8_t * mask, int mask_stride, int mode, int swap)
{
    if(mode < 1 || mode > 3) return 0; /* "unsupported blend mode": warning + no-op (:556-558, :900-902) */
    for(int y = 0; y < h; y++) {
        uint16_t * d = dest + (size_t)y * dest_stride;
        const uint16_t * s = src ? src + (size_t)y * src_stride : NULL;
        const uint8_t * m = mask ? mask + (size_t)y * mask_stride : NULL;
        for(int x = 0; x < w; x++) {
            uint8_t o = opa;
            if(m) {
                if(m[x] == 0) continue;
                o = m[x] >= OPA_MAX ? opa : (uint8_t)(((uint32_t)m[x] * opa) >> 8);
            }
            d[x] = blend_mode_px(mode, s ? s[x] : color, d[x], o, swap);
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------------------------ */
/* ARGB8565 destination (screen_transp), libraries/lvgl/src/draw/sw/lv_draw_sw_blend.c:378-530, 737-874         */
/* ------------------------------------------------------------------------------------------------------------ */
static void set_px_argb(uint8_t * p, uint16_t color, uint8_t opa, int swap) /* :378-402 */
{
    uint16_t bg = (uint16_t)(p[0] + (p[1] << 8));
    uint16_t res;
    uint8_t ro;
    lvref_color_mix_with_alpha(bg, p[2], color, opa, &res, &ro, swap);
    p[2] = ro;
    if(ro <= OPA_MIN) return;
    p[0] = (uint8_t)(res & 0xFF);
    p[1] = (uint8_t)(res >> 8);
}

static void set_px_argb_blend(uint8_t * p, uint16_t color, uint8_t opa, int mode, int swap) /* :404-446 */
{
    if(p[2] <= OPA_MIN) return;
    uint16_t bg = (uint16_t)(p[0] + (p[1] << 8));
    uint16_t res = blend_mode_px(mode, color, bg, opa, swap);
    p[0] = (uint8_t)(res & 0xFF);
    p[1] = (uint8_t)(res >> 8);
}

static void fill_argb(uint8_t * dest, int w, int h, int dest_stride, uint16_t color, uint8_t opa,
                      const uint8_t * mask, int mask_stride, int swap) /* :448-530 */
{
    for(int y = 0; y < h; y++) {
        uint8_t * d = dest + (size_t)y * dest_stride * 3;
        for(int x = 0; x < w; x++, d += 3) {
            if(mask == NULL) {
                if(opa >= OPA_MAX) { /* :465-477 */
                    d[0] = (uint8_t)(color & 0xFF);
                    d[1] = (uint8_t)(color >> 8);
                    d[2] = opa;
                }
                else set_px_argb(d, color, opa, swap); /* :479-489 */
            }
            else if(opa >= OPA_MAX) {
                /* :494-505: the mask pointer is only ever incremented per pixel there -- it is never advanced by
                 * (mask_stride - w) at the end of a row, so row y starts at mask[y*w], not mask[y*mask_stride]. */
                set_px_argb(d, color, mask[(size_t)y * w + x], swap);
            }
            else { /* :507-528 */
                uint8_t m = mask[(size_t)y * mask_stride + x];
                if(m) {
                    uint8_t o = m == OPA_COVER ? opa : (uint8_t)(((uint32_t)m * opa) >> 8);
                    set_px_argb(d, color, o, swap);
                }
            }
        }
    }
}

static void map_argb(uint8_t * dest, int w, int h, int dest_stride, const uint16_t * src, int src_stride, uint8_t opa,
                     const uint8_t * mask, int mask_stride, int mode, int swap) /* :737-874 */
{
    int blended = mode >= 1 && mode <= 3; /* any other mode: blend_fp == NULL -> set_px_argb (:762-763) */
    for(int y = 0; y < h; y++) {
        uint8_t * d = dest + (size_t)y * dest_stride * 3;
        const uint16_t * s = src + (size_t)y * src_stride;
        const uint8_t * m = mask ? mask + (size_t)y * mask_stride : NULL;
        for(int x = 0; x < w; x++, d += 3) {
            uint8_t o;
            if(m == NULL) o = opa >= OPA_MAX ? OPA_COVER : opa; /* :768-819 */
            else if(opa > OPA_MAX) o = m[x];                    /* :824-843: mask value 0 is passed through */
            else {                                              /* :846-871 */
                if(m[x] == 0) continue;
                o = m[x] >= OPA_MAX ? opa : (uint8_t)(((uint32_t)opa * m[x]) >> 8);
            }
            if(blended) set_px_argb_blend(d, s[x], o, mode, swap);
            else set_px_argb(d, s[x], o, swap);
        }
    }
}

/* lv_draw_sw_blend :95-106 + lv_draw_sw_blend_basic :108-204 (set_px_cb == NULL). */
int lvref_blend(const lvref_blend_t * b)
{
    if(b->opa <= OPA_MIN) return 0; /* :98 */
    lvref_area_t a;
    if(!lvref_area_intersect(&a, &b->blend_area, &b->clip_area)) return 0; /* :101, :120 */

    const uint8_t * mask;
    if(b->mask_buf && b->mask_res == 0 /*TRANSP*/) return 0; /* :113 */
    else if(b->mask_res == 1 /*FULL_COVER*/) mask = NULL;   /* :114 */
    else mask = b->mask_buf;                                /* :112, :115 */

    int dest_stride = aw(&b->buf_area); /* :117 */
    size_t dest_ofs = (size_t)dest_stride * (a.y1 - b->buf_area.y1) + (a.x1 - b->buf_area.x1); /* :126 */

    const uint16_t * src = b->src_buf;
    int src_stride = 0;
    if(src) {
        src_stride = aw(&b->blend_area); /* :140-141 */
        src += (size_t)src_stride * (a.y1 - b->blend_area.y1) + (a.x1 - b->blend_area.x1);
    }
    int mask_stride = 0;
    if(mask) {
        if(b->antialiasing_off) { /* :150-156: the caller's mask itself is rounded, all of mask_area */
            uint8_t * m = (uint8_t *)b->mask_buf;
            int32_t mask_size = (int32_t)aw(&b->mask_area) * ah(&b->mask_area);
            for(int32_t i = 0; i < mask_size; i++) m[i] = m[i] > 128 ? 255 : 0;
        }
        mask_stride = aw(&b->mask_area); /* :158-159 */
        mask += (size_t)mask_stride * (a.y1 - b->mask_area.y1) + (a.x1 - b->mask_area.x1);
    }
    int w = aw(&a), h = ah(&a);

    if(b->screen_transp) { /* :177-184, destination pixels are 3 bytes (:129-133) */
        uint8_t * d8 = (uint8_t *)b->buf + dest_ofs * 3;
        if(src == NULL) fill_argb(d8, w, h, dest_stride, b->color, b->opa, mask, mask_stride, b->swap);
        else map_argb(d8, w, h, dest_stride, src, src_stride, b->opa, mask, mask_stride, b->blend_mode, b->swap);
        return 0;
    }
    uint16_t * d = (uint16_t *)b->buf + dest_ofs;
    if(b->blend_mode == 0) { /* :186-193 */
        if(src == NULL) fill_normal(d, w, h, dest_stride, b->color, b->opa, mask, mask_stride, b->swap);
        else map_normal(d, w, h, dest_stride, src, src_stride, b->opa, mask, mask_stride, b->swap);
        return 0;
    }
    return fill_or_map_blended(d, w, h, dest_stride, src, src_stride, b->color, b->opa, mask, mask_stride,
                               b->blend_mode, b->swap); /* :194-203 */
}

/* ------------------------------------------------------------------------------------------------------------ */
/* Image colour/alpha split                                                                                     */
/* ------------------------------------------------------------------------------------------------------------ */
void lvref_image_split(const void * src, int format, uint16_t * color, uint8_t * alpha, size_t n, int swap)
{
    const uint8_t * p = (const uint8_t *)src;
    for(size_t i = 0; i < n; i++) {
        if(format == 2) { /* lv_color32_t {blue, green, red, alpha} -> lv_color_make(r,g,b); lv_png.c:244-253 */
            color[i] = lvref_color_make(p[4 * i + 2], p[4 * i + 1], p[4 * i + 0], swap);
            alpha[i] = p[4 * i + 3];
        }
        else if(format == 3) { /* lodepng byte order R,G,B,A: lv_png.c:249 reads .blue/.green/.red as r/g/b */
            color[i] = lvref_color_make(p[4 * i + 0], p[4 * i + 1], p[4 * i + 2], swap);
            alpha[i] = p[4 * i + 3];
        }
        else { /* 3 bytes per pixel; lv_draw_sw_img.c:259-263 */
            color[i] = (uint16_t)(p[3 * i] + (p[3 * i + 1] << 8));
            alpha[i] = p[3 * i + 2];
        }
    }
}

/* What lv_draw_sw_img_decoded puts into its colour and alpha buffers for n pixels of an image before it blends them:
 * convert_cb (libraries/lvgl/src/draw/sw/lv_draw_sw_img.c:203-298) -- format 1 = LV_IMG_CF_TRUE_COLOR (alpha stays
 * 0xFF, :215-216), with chroma_keyed != 0 LV_IMG_CF_TRUE_COLOR_CHROMA_KEYED (pixels equal to the key get alpha 0,
 * :230-247); formats 2/3/4 as lvref_image_split -- followed by the recolor pass (:163-174): every colour becomes
 * lv_color_mix_premult(premult(recolor, recolor_opa), colour, 255 - recolor_opa) when recolor_opa > LV_OPA_MIN. */
void lvref_image_convert(const void * src, int format, uint16_t * color, uint8_t * alpha, size_t n, int swap,
                         int chroma_keyed, uint16_t chroma_key, uint16_t recolor, uint8_t recolor_opa)
{
    if(format == 1) {
        const uint16_t * s = (const uint16_t *)src;
        for(size_t i = 0; i < n; i++) {
            color[i] = s[i];
            alpha[i] = (chroma_keyed && s[i] == chroma_key) ? 0x00 : 0xFF;
        }
    }
    else lvref_image_split(src, format, color, alpha, n, swap);
    if(recolor_opa > OPA_MIN) {
        uint16_t pm[3];
        lvref_color_premult(recolor, recolor_opa, pm, swap);
        uint8_t inv = (uint8_t)(255 - recolor_opa);
        for(size_t i = 0; i < n; i++) color[i] = lvref_color_mix_premult(pm, color[i], inv, swap);
    }
}

/* libraries/lvgl/src/draw/sw/lv_draw_sw_img.c:98-195 with no transform, no recolor, no extra draw masks. */
int lvref_img_blend(void * buf, lvref_area_t buf_area, lvref_area_t clip_area, lvref_area_t coords, const void * src,
                    int format, uint8_t opa, int swap, int hor_res)
{
    return lvref_img_blend_aa(buf, buf_area, clip_area, coords, src, format, opa, swap, hor_res, 0);
}

int lvref_img_blend_aa(void * buf, lvref_area_t buf_area, lvref_area_t clip_area, lvref_area_t coords, const void * src,
                       int format, uint8_t opa, int swap, int hor_res, int antialiasing_off)
{
    lvref_area_t blend_area;
    if(!lvref_area_intersect(&blend_area, &clip_area, &coords)) return 0; /* callers pass clip within coords */
    int src_w = aw(&coords);
    int blend_w = aw(&blend_area), blend_h = ah(&blend_area);
    int bpp = format == 4 ? 3 : 4;

    uint32_t max_buf_size = (uint32_t)hor_res; /* MAX_BUF_SIZE, :20 */
    uint32_t blend_size = (uint32_t)blend_w * (uint32_t)blend_h;
    uint32_t buf_h = blend_size <= max_buf_size ? (uint32_t)blend_h : max_buf_size / (uint32_t)blend_w; /* :113-119 */
    if(buf_h == 0) return -1;
    size_t buf_size = (size_t)blend_w * buf_h;
    uint16_t * rgb = malloc(buf_size * 2);
    uint8_t * a8 = malloc(buf_size);

    int y_last = blend_area.y2;
    lvref_area_t chunk = blend_area;
    chunk.y2 = (int16_t)(chunk.y1 + (int)buf_h - 1);
    if(chunk.y2 > y_last) chunk.y2 = (int16_t)y_last;
    while(chunk.y1 <= y_last) { /* :138-191 */
        int ch_h = ah(&chunk);
        for(int y = 0; y < ch_h; y++) { /* convert_cb, :249-275, rows of the chunk relative to the image */
            size_t sy = (size_t)(chunk.y1 - coords.y1 + y);
            size_t sx = (size_t)(chunk.x1 - coords.x1);
            const uint8_t * sp = (const uint8_t *)src + (sy * (size_t)src_w + sx) * (size_t)bpp;
            lvref_image_split(sp, format, rgb + (size_t)y * blend_w, a8 + (size_t)y * blend_w, (size_t)blend_w, swap);
        }
        lvref_blend_t b;
        memset(&b, 0, sizeof(b));
        b.buf = buf;
        b.buf_area = buf_area;
        b.clip_area = clip_area;
        b.blend_area = chunk;
        b.src_buf = rgb;
        b.mask_buf = a8;
        b.mask_res = 2; /* CHANGED, :133-136 */
        b.mask_area = chunk;
        b.opa = opa;
        b.swap = (uint8_t)swap;
        b.antialiasing_off = antialiasing_off ? 1 : 0;
        lvref_blend(&b); /* :185 */
        chunk.y1 = (int16_t)(chunk.y2 + 1);
        chunk.y2 = (int16_t)(chunk.y1 + (int)buf_h - 1);
        if(chunk.y2 > y_last) chunk.y2 = (int16_t)y_last;
    }
    free(rgb);
    free(a8);
    return 0;
}

/* ------------------------------------------------------------------------------------------------------------ */
/* lvgl_port_flush_callback, src/lv_port.c:428-572                                                              */
/* ------------------------------------------------------------------------------------------------------------ */
int lvref_flush(const lvref_area_t * area, const uint16_t * color_map, int rotation, uint32_t trans_size, int hor_res,
                int ver_res, lvref_draw_bitmap_cb cb, void * user)
{
    const int x_start = area->x1, x_end = area->x2, y_start = area->y1, y_end = area->y2;
    const int width = x_end - x_start + 1, height = y_end - y_start + 1;
    const uint16_t * from = color_map;

    if(trans_size == 0) { /* :568-570 */
        cb(user, x_start, y_start, x_end + 1, y_end + 1, color_map);
        return 1;
    }
    uint16_t * bufs[2];
    bufs[0] = malloc((size_t)trans_size * 2);
    bufs[1] = malloc((size_t)trans_size * 2);
    int act = 0; /* trans_act starts at trans_buf_1 (:453) and is toggled BEFORE each use (:496) */

    int x_start_tmp = 0, x_end_tmp = 0, max_width = 0, trans_width = 0;
    int y_start_tmp = 0, y_end_tmp = 0, max_height = 0, trans_height = 0;
    int trans_count;
    int xd0 = 0, xd1 = 0, yd0 = 0, yd1 = 0;
    const int transposing = rotation == 1 || rotation == 3;

    if(transposing) { /* :466-471 */
        int per = (int)(trans_size / (uint32_t)height);
        max_width = per > width ? width : per;
        trans_count = width / max_width + (width % max_width ? 1 : 0);
        x_start_tmp = x_start;
        x_end_tmp = x_end;
    }
    else { /* :472-478 */
        int per = (int)(trans_size / (uint32_t)width);
        max_height = per > height ? height : per;
        trans_count = height / max_height + (height % max_height ? 1 : 0);
        y_start_tmp = y_start;
        y_end_tmp = y_end;
    }

    for(int i = 0; i < trans_count; i++) {
        if(rotation == 1) { /* :482-484 */
            int left = x_end - x_start_tmp + 1;
            trans_width = left > max_width ? max_width : left;
            x_end_tmp = left > max_width ? x_start_tmp + max_width - 1 : x_end;
        }
        else if(rotation == 3) { /* :485-487 */
            int left = x_end_tmp - x_start + 1;
            trans_width = left > max_width ? max_width : left;
            x_start_tmp = left > max_width ? x_end_tmp - trans_width + 1 : x_start;
        }
        else if(rotation == 0) { /* :488-490 */
            int left = y_end - y_start_tmp + 1;
            trans_height = left > max_height ? max_height : left;
            y_end_tmp = left > max_height ? y_start_tmp + max_height - 1 : y_end;
        }
        else { /* :491-494 */
            int left = y_end_tmp - y_start + 1;
            trans_height = left > max_height ? max_height : left;
            y_start_tmp = left > max_height ? y_end_tmp - max_height + 1 : y_start;
        }
        act ^= 1;
        uint16_t * to = bufs[act];

        switch(rotation) {
            case 1: /* :500-510 */
                for(int y = 0; y < height; y++)
                    for(int x = 0; x < trans_width; x++)
                        to[x * height + (height - y - 1)] = from[y * width + x_start_tmp + x];
                xd0 = ver_res - y_end - 1;
                xd1 = ver_res - y_start - 1;
                yd0 = x_start_tmp;
                yd1 = x_end_tmp;
                break;
            case 3: /* :511-521 */
                for(int y = 0; y < height; y++)
                    for(int x = 0; x < trans_width; x++)
                        to[(trans_width - x - 1) * height + y] = from[y * width + x_start_tmp + x];
                xd0 = y_start;
                xd1 = y_end;
                yd0 = hor_res - x_end_tmp - 1;
                yd1 = hor_res - x_start_tmp - 1;
                break;
            case 2: /* :522-532 */
                for(int y = 0; y < trans_height; y++)
                    for(int x = 0; x < width; x++)
                        to[(trans_height - y - 1) * width + (width - x - 1)] = from[y_start_tmp * width + y * width + x];
                xd0 = hor_res - x_end - 1;
                xd1 = hor_res - x_start - 1;
                yd0 = ver_res - y_end_tmp - 1;
                yd1 = ver_res - y_start_tmp - 1;
                break;
            default: /* :533-543 */
                for(int y = 0; y < trans_height; y++)
                    for(int x = 0; x < width; x++) to[y * width + x] = from[y_start_tmp * width + y * width + x];
                xd0 = x_start;
                xd1 = x_end;
                yd0 = y_start_tmp;
                yd1 = y_end_tmp;
                break;
        }
        cb(user, xd0, yd0, xd1 + 1, yd1 + 1, to); /* :556 */

        /* :558-566 (the missing 'else' before the ROT_NONE test there only touches unused variables) */
        if(rotation == 1) x_start_tmp += max_width;
        else if(rotation == 3) x_end_tmp -= max_width;
        if(rotation == 0) y_start_tmp += max_height;
        else y_end_tmp -= max_height;
    }
    free(bufs[0]);
    free(bufs[1]);
    return trans_count;
}

void lvref_flush_frame(const uint16_t * in, uint16_t * out, int W, int H, int rotation, int swap_bytes)
{
    for(int y = 0; y < H; y++) {
        for(int x = 0; x < W; x++) {
            uint16_t v = in[(size_t)y * W + x];
            if(swap_bytes) v = bswap16(v);
            size_t o;
            switch(rotation) {
                case 1: o = (size_t)x * H + (size_t)(H - 1 - y); break;
                case 2: o = (size_t)(H - 1 - y) * W + (size_t)(W - 1 - x); break;
                case 3: o = (size_t)(W - 1 - x) * H + (size_t)y; break;
                default: o = (size_t)y * W + x; break;
            }
            out[o] = v;
        }
    }
}

/* ------------------------------------------------------------------------------------------------------------ */
/* Panel driver: panel_axs15231b_draw_bitmap, src/esp_lcd_axs15231b.c:292-337                                    */
/* ------------------------------------------------------------------------------------------------------------ */
static void panel_tx_param(int qspi, lvref_panel_tx_cb tx, void * user, int lcd_cmd, const uint8_t * param, size_t n)
{
    if(qspi) { /* :176-180 */
        lcd_cmd &= 0xff;
        lcd_cmd <<= 8;
        lcd_cmd |= 0x02 << 24; /* LCD_OPCODE_WRITE_CMD, :27 */
    }
    tx(user, 0, lcd_cmd, param, n);
}
static void panel_tx_color(int qspi, lvref_panel_tx_cb tx, void * user, int lcd_cmd, size_t n)
{
    if(qspi) { /* :187-191 */
        lcd_cmd &= 0xff;
        lcd_cmd <<= 8;
        lcd_cmd |= 0x32 << 24; /* LCD_OPCODE_WRITE_COLOR, :29 */
    }
    tx(user, 1, lcd_cmd, NULL, n);
}
void lvref_axs15231b_draw_bitmap(int x_start, int y_start, int x_end, int y_end, int x_gap, int y_gap, int qspi,
                                 int fb_bits_per_pixel, lvref_panel_tx_cb tx, void * user)
{
    uint8_t p[4];
    x_start += x_gap; /* :301-304 */
    x_end += x_gap;
    y_start += y_gap;
    y_end += y_gap;
    p[0] = (x_start >> 8) & 0xFF; /* :307-314 */
    p[1] = x_start & 0xFF;
    p[2] = ((x_end - 1) >> 8) & 0xFF;
    p[3] = (x_end - 1) & 0xFF;
    panel_tx_param(qspi, tx, user, 0x2A /* LCD_CMD_CASET */, p, 4);
    if(!qspi) { /* :316-325 */
        p[0] = (y_start >> 8) & 0xFF;
        p[1] = y_start & 0xFF;
        p[2] = ((y_end - 1) >> 8) & 0xFF;
        p[3] = (y_end - 1) & 0xFF;
        panel_tx_param(qspi, tx, user, 0x2B /* LCD_CMD_RASET */, p, 4);
    }
    size_t len = (size_t)(x_end - x_start) * (y_end - y_start) * fb_bits_per_pixel / 8; /* :328-329 */
    if(y_start == 0) panel_tx_color(qspi, tx, user, 0x2C /* LCD_CMD_RAMWR */, len);   /* :330-331 */
    else panel_tx_color(qspi, tx, user, 0x3C /* LCD_CMD_RAMWRC */, len);              /* :332-333 */
}

/* ------------------------------------------------------------------------------------------------------------ */
/* Synthetic inputs and checksum                                                                                */
/* ------------------------------------------------------------------------------------------------------------ */
static inline uint64_t fin64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static inline uint64_t synth_hash(uint64_t seed, uint64_t idx)
{
    return fin64(seed + (idx + 1) * 0x9E3779B97F4A7C15ULL);
}
static inline uint8_t ui_alpha(uint64_t h)
{
    uint32_t t = (uint32_t)(h >> 56) % 3u;
    if(t == 0) return 0;
    if(t == 1) return 255;
    return (uint8_t)(1u + ((uint32_t)(h >> 32) & 0xFFFFu) % 254u);
}

void lvref_synth(void * out, int kind, uint64_t seed, uint64_t first_index, size_t count)
{
    for(size_t i = 0; i < count; i++) {
        uint64_t h = synth_hash(seed, first_index + i);
        switch(kind) {
            case 0: ((uint16_t *)out)[i] = (uint16_t)(h & 0xFFFF); break;
            case 1: ((uint8_t *)out)[i] = ui_alpha(h); break;
            case 2: ((uint8_t *)out)[i] = (uint8_t)(h & 0xFF); break;
            case 3: ((uint32_t *)out)[i] = (uint32_t)(h & 0xFFFFFF) | ((uint32_t)ui_alpha(h) << 24); break;
            default: ((uint32_t *)out)[i] = (uint32_t)h; break;
        }
    }
}

uint64_t lvref_checksum(const void * in, int elem_bytes, uint64_t first_index, size_t count)
{
    uint64_t s = 0;
    for(size_t i = 0; i < count; i++) {
        uint64_t v = elem_bytes == 1 ? ((const uint8_t *)in)[i]
                     : elem_bytes == 2 ? ((const uint16_t *)in)[i]
                                       : ((const uint32_t *)in)[i];
        s += fin64((first_index + i) * 0x9E3779B97F4A7C15ULL + v * 0xD1B54A32D192ED03ULL + 1);
    }
    return s;
}

/* ------------------------------------------------------------------------------------------------------------ */
/* CPU baseline driver: frames are independent, one thread per frame range.                                     */
/* ------------------------------------------------------------------------------------------------------------ */
struct cat_ctx {
    uint16_t * out;
    size_t pos;
};
static void cat_cb(void * user, int x0, int y0, int x1, int y1, const uint16_t * data)
{
    struct cat_ctx * c = (struct cat_ctx *)user;
    size_t n = (size_t)(x1 - x0) * (size_t)(y1 - y0);
    memcpy(c->out + c->pos, data, n * 2);
    c->pos += n;
}

void lvref_argb_blend_flush_batch(uint16_t * dst, const uint32_t * src, uint16_t * out, int width, int height,
                                  int n_frames, int rotation, int swap, int threads)
{
    lvref_area_t full = {0, 0, (int16_t)(width - 1), (int16_t)(height - 1)};
    size_t px = (size_t)width * height;
    uint32_t trans_size = (uint32_t)(px / 10); /* src/esp_bsp.c:334 */
    int hres = width, vres = height;           /* drv->hor_res / ver_res = the logical LVGL resolution */
#ifdef _OPENMP
    if(threads > 0) omp_set_num_threads(threads);
#pragma omp parallel for schedule(static)
#endif
    for(int f = 0; f < n_frames; f++) {
        uint16_t * d = dst + (size_t)f * px;
        lvref_img_blend(d, full, full, full, src + (size_t)f * px, 2, 255, swap, hres);
        if(out) {
            struct cat_ctx c = {out + (size_t)f * px, 0};
            lvref_flush(&full, d, rotation, trans_size, hres, vres, cat_cb, &c);
        }
    }
    (void)threads;
}

/* A display list -- n_items descriptors applied in order -- replayed over n_frames draw buffers of width x height
 * (OpenMP over frames), each followed by the flush rotation when out != NULL.  The descriptors' buf field is ignored;
 * their sources and masks are shared by all frames.  This is the CPU side of the bench's scene workload. */
void lvref_list_flush_batch(uint16_t * dst, const lvref_blend_t * items, int n_items, uint16_t * out, int width, int height,
                            int n_frames, int rotation, int threads)
{
    lvref_area_t full = {0, 0, (int16_t)(width - 1), (int16_t)(height - 1)};
    size_t px = (size_t)width * height;
    uint32_t trans_size = (uint32_t)(px / 10);
#ifdef _OPENMP
    if(threads > 0) omp_set_num_threads(threads);
#pragma omp parallel for schedule(static)
#endif
    for(int f = 0; f < n_frames; f++) {
        uint16_t * d = dst + (size_t)f * px;
        for(int i = 0; i < n_items; i++) {
            lvref_blend_t b = items[i];
            b.buf = d;
            lvref_blend(&b);
        }
        if(out) {
            struct cat_ctx c = {out + (size_t)f * px, 0};
            lvref_flush(&full, d, rotation, trans_size, width, height, cat_cb, &c);
        }
    }
    (void)threads;
}
