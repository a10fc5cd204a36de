/*
 * lvgpu_lvgl.h -- the LVGL-signature plug-in callbacks of liblvgpu.so.
 *
 * These are the functions an LVGL 8.x port installs between lv_disp_drv_init() and lv_disp_drv_register()
 * (libraries/lvgl/docs/porting/gpu.md:80-161; precedent libraries/lvgl/src/draw/stm32_dma2d/lv_gpu_stm32_dma2d.c:107-117):
 *
 *     lv_disp_drv_init(&drv);
 *     drv.draw_ctx_init   = lvgpu_draw_ctx_init;
 *     drv.draw_ctx_deinit = lvgpu_draw_ctx_deinit;
 *     drv.draw_ctx_size   = lvgpu_draw_ctx_size();
 *     drv.flush_cb        = lvgpu_flush_cb;            // optional: replaces lvgl_port_flush_callback
 *     drv.user_data       = &my_lvgpu_flush_cfg;       // only read by lvgpu_flush_cb
 *
 * The library is built WITHOUT LVGL headers (the GPU box has none), so the callbacks are declared on mirror structs
 * that are layout-compatible with LVGL 8.3/8.4 on a 64-bit little-endian host with LV_COLOR_DEPTH 16,
 * LV_USE_USER_DATA 1 and LV_USE_LARGE_COORD 0 (the reference firmware's src/lv_conf.h:27,307,356).  A translation unit
 * that has already included lvgl.h gets the same prototypes on LVGL's own types instead.
 *
 * These single-frame callbacks take HOST buffers (LVGL's draw buffer lives in host memory): every call stages the
 * touched rectangle to the device, runs the same kernels as the batched entry points of lvgpu.h and copies the
 * result back before returning, because the reference lets callers reuse src_buf/mask_buf as soon as blend()
 * returns (libraries/lvgl/src/draw/sw/lv_draw_sw_rect.c:224-232) and reads the draw buffer on the CPU.
 */
#ifndef LVGPU_LVGL_H
#define LVGPU_LVGL_H

#include "lvgpu.h"

#ifdef __cplusplus
extern "C" {
#endif

#ifdef LVGL_H
typedef lv_area_t lvgpu_lv_area_t;
typedef lv_draw_ctx_t lvgpu_lv_draw_ctx_t;
typedef lv_draw_sw_blend_dsc_t lvgpu_lv_draw_sw_blend_dsc_t;
typedef lv_disp_drv_t lvgpu_lv_disp_drv_t;
typedef lv_color_t lvgpu_lv_color_t;
#else
typedef lvgpu_area_t lvgpu_lv_area_t;
typedef uint16_t lvgpu_lv_color_t; /* lv_color_t.full, libraries/lvgl/src/misc/lv_color.h:208-222 */

/* lv_draw_ctx_t, libraries/lvgl/src/draw/lv_draw.h:59-189 (168 bytes) */
typedef struct lvgpu_lv_draw_ctx_t {
    void * buf;
    lvgpu_lv_area_t * buf_area;
    const lvgpu_lv_area_t * clip_area;
    void (*init_buf)(struct lvgpu_lv_draw_ctx_t *);
    void * draw_rect;
    void * draw_arc;
    void * draw_img_decoded;
    void * draw_img;
    void * draw_letter;
    void * draw_line;
    void * draw_polygon;
    void * draw_transform;
    void * draw_bg;
    void (*wait_for_finish)(struct lvgpu_lv_draw_ctx_t *);
    void (*buffer_copy)(struct lvgpu_lv_draw_ctx_t *, void * dest_buf, int16_t dest_stride,
                        const lvgpu_lv_area_t * dest_area, void * src_buf, int16_t src_stride,
                        const lvgpu_lv_area_t * src_area);
    void * layer_init;
    void * layer_adjust;
    void * layer_blend;
    void * layer_destroy;
    size_t layer_instance_size;
    void * user_data;
} lvgpu_lv_draw_ctx_t;

/* lv_draw_sw_blend_dsc_t, libraries/lvgl/src/draw/sw/lv_draw_sw_blend.h:29-39 (56 bytes) */
typedef struct {
    const lvgpu_lv_area_t * blend_area;
    const lvgpu_lv_color_t * src_buf;
    lvgpu_lv_color_t color;
    uint8_t * mask_buf;
    uint8_t mask_res;
    const lvgpu_lv_area_t * mask_area;
    uint8_t opa;
    uint8_t blend_mode;
} lvgpu_lv_draw_sw_blend_dsc_t;

/* lv_disp_draw_buf_t, libraries/lvgl/src/hal/lv_hal_disp.h:51-64 (40 bytes) */
typedef struct {
    void * buf1;
    void * buf2;
    void * buf_act;
    uint32_t size;
    int flushing;
    int flushing_last;
    uint32_t last_area : 1;
    uint32_t last_part : 1;
} lvgpu_lv_disp_draw_buf_t;

/* lv_disp_drv_t, libraries/lvgl/src/hal/lv_hal_disp.h:78-152 (152 bytes) */
typedef struct lvgpu_lv_disp_drv_t {
    int16_t hor_res, ver_res, physical_hor_res, physical_ver_res, offset_x, offset_y;
    lvgpu_lv_disp_draw_buf_t * draw_buf;
    uint32_t direct_mode : 1;
    uint32_t full_refresh : 1;
    uint32_t sw_rotate : 1;
    uint32_t antialiasing : 1;
    uint32_t rotated : 2;
    uint32_t screen_transp : 1;
    uint32_t dpi : 10;
    void (*flush_cb)(struct lvgpu_lv_disp_drv_t *, const lvgpu_lv_area_t *, lvgpu_lv_color_t *);
    void * rounder_cb;
    void * set_px_cb;
    void * clear_cb;
    void * monitor_cb;
    void * wait_cb;
    void * clean_dcache_cb;
    void * drv_update_cb;
    void * render_start_cb;
    lvgpu_lv_color_t color_chroma_key;
    lvgpu_lv_draw_ctx_t * draw_ctx;
    void (*draw_ctx_init)(struct lvgpu_lv_disp_drv_t *, lvgpu_lv_draw_ctx_t *);
    void (*draw_ctx_deinit)(struct lvgpu_lv_disp_drv_t *, lvgpu_lv_draw_ctx_t *);
    size_t draw_ctx_size;
    void * user_data;
} lvgpu_lv_disp_drv_t;
#endif /* LVGL_H */

/*
 * Tell the library where the host program's LVGL lives.  Needed because the replacement context, like every LVGL
 * GPU back-end, starts from the software context (lv_draw_sw_init_ctx, libraries/lvgl/src/draw/sw/lv_draw_sw.c:40-65)
 * and hands requests it does not cover to lv_draw_sw_blend_basic (libraries/lvgl/src/draw/sw/lv_draw_sw_blend.c:108).
 * Any argument may be NULL: the library then looks the symbol up with dlsym(RTLD_DEFAULT, ...), which finds it when
 * LVGL is linked into the program normally.  lv_disp_flush_ready_fn == NULL and not found: the library clears
 * draw_buf->flushing / flushing_last itself (libraries/lvgl/src/hal/lv_hal_disp.c:519-522).
 */
void lvgpu_lvgl_bind(void * lv_draw_sw_init_ctx_fn, void * lv_draw_sw_blend_basic_fn, void * lv_disp_flush_ready_fn);

/* LV_COLOR_16_SWAP of the host's LVGL build (default 1, the reference firmware's setting, src/lv_conf.h:30). */
void lvgpu_lvgl_set_color_16_swap(int swap);

/*
 * Deferred mode (read by lvgpu_draw_ctx_init).  on = 1: on; 0: off (immediate mode: every blend call is a round trip
 * to the device, correct anywhere and two orders of magnitude slower than LVGL on a live display); -1, the default:
 * on for a display whose flush_cb is lvgpu_flush_cb -- the usual installation, in which the flush callback
 * draws what is queued before it looks at the buffer -- and off for a display that keeps a flush_cb of its own, which
 * has to opt in because it has to cooperate (below).  LVGL's producers issue one lv_draw_sw_blend per row of
 * a rounded, anti-aliased or text object (libraries/lvgl/src/draw/sw/lv_draw_sw_rect.c:118-330, lv_draw_sw_letter.c:
 * 246-398) -- a 480x320 widget screen is 800-1600 calls.  With deferred mode on, lvgpu_blend only copies the call's
 * source and mask rectangles into a queue; the queue is drawn as ONE display list (lvgpu_blend_list_batch) when the
 * buffer is about to be looked at: in lvgpu_flush_cb, lvgpu_wait_for_finish, the layer_* and buffer_copy hooks of
 * the context (wrapped for this purpose), when draw_ctx->buf changes, and in lvgpu_draw_ctx_deinit.
 * draw_ctx->wait_for_finish itself is left NULL in this mode, because lv_draw_sw_blend calls it before every single
 * blend (libraries/lvgl/src/draw/sw/lv_draw_sw_blend.c:103); all of LVGL's callers check the pointer.  A display
 * driver that keeps its own flush_cb must therefore call lvgpu_wait_for_finish(drv->draw_ctx) first thing in it.
 * Host code that reads or writes draw_ctx->buf directly between blends, other than through the hooks above, is not
 * supported in this mode: it would see pixels that are still queued.  LVGL itself has one such reader, the sub-pixel
 * text renderer (draw_letter_subpx reads the destination, libraries/lvgl/src/draw/sw/lv_draw_sw_letter.c:423), so
 * the mode requires LV_USE_FONT_SUBPX 0 (the reference firmware's setting).  These two requirements -- a cooperating
 * flush_cb and no direct readers -- are why the mode is opt-in.
 * A draw buffer that came from lvgpu_lvgl_buf_alloc (below) is drawn IN PLACE: the launch reads sources and masks from
 * the queue's pinned memory and reads and writes the draw buffer (and, from lvgpu_flush_cb, writes the panel image) in
 * host memory over the bus, so a submission is one launch and one wait (lvgpu_lvgl_stats_t.lists_in_place).  Any other
 * buffer (LVGL's own layer buffers, a display buffer from malloc) is staged: it travels to the device only when no
 * queued item overwrites all of it (a screen background does), and comes back once per submission;
 * LVGPU_LVGL_STAGED in the environment stages everything.  A refresh that queues exactly what the previous one queued (same rectangles,
 * colours, opacities; mask and image bytes may differ) replays the plan made for the previous one instead of planning
 * again (lvgpu_lvgl_stats_t.list_replays).  A display with drv->sw_rotate set (LVGL rotates the draw buffer on the CPU right before flush_cb,
 * libraries/lvgl/src/core/lv_refr.c:1262-1296, with no hook in between) stays in immediate mode.
 */
void lvgpu_lvgl_set_deferred(int on);

/*
 * What the callbacks do when the GPU fails them (they return void, LVGL has no error channel).  In immediate mode a
 * failure before the blended rectangle starts to travel back -- no memory for the staging buffers, a refused copy or
 * launch -- leaves the host buffer untouched: it is counted (lvgpu_lvgl_stats_t.errors and blend_delegated), reported on
 * stderr the first few times, and the call is drawn by the host program's lv_draw_sw_blend_basic, the convention of
 * LVGL's GPU back-ends (libraries/lvgl/src/draw/stm32_dma2d/lv_gpu_stm32_dma2d.c:127-130); the program keeps running.
 * With no lv_draw_sw_blend_basic bound, in deferred mode (the calls that filled the queue have long returned and their
 * sources are gone) and once a result is on its way back, a failure ends the program with a message: a frame with a
 * hole in it must not go unnoticed.  lvgpu_lvgl_inject_failure(n) makes the n-th immediate blend from now fail before
 * it touches the GPU (0 = none): a hook for tests of the above.
 */
void lvgpu_lvgl_inject_failure(int nth_blend);

/*
 * Memory for the display's draw buffers (the argument of lv_disp_draw_buf_init): what the port takes from
 * heap_caps_malloc with MALLOC_CAP_DMA when the board sets flags.buff_dma (src/lv_port.c:198-207) and releases with
 * free (src/lv_port.c:256-258, 286-288).  Here "DMA-capable" means pinned host memory that the GPU reads and writes
 * directly: deferred mode draws such a buffer in place, without copying it to the device and back.  For the CPU it is
 * ordinary cached memory.  NULL when the allocation fails.  A buffer from anywhere else works too, through copies.
 */
void * lvgpu_lvgl_buf_alloc(size_t bytes);
void lvgpu_lvgl_buf_free(void * p);

/* drv->draw_ctx_init / draw_ctx_deinit / draw_ctx_size (libraries/lvgl/src/hal/lv_hal_disp.h:143-146).
 * init = lv_draw_sw_init_ctx, then blend and wait_for_finish are replaced. */
void lvgpu_draw_ctx_init(lvgpu_lv_disp_drv_t * drv, lvgpu_lv_draw_ctx_t * draw_ctx);
void lvgpu_draw_ctx_deinit(lvgpu_lv_disp_drv_t * drv, lvgpu_lv_draw_ctx_t * draw_ctx);
size_t lvgpu_draw_ctx_size(void);

/* lv_draw_sw_ctx_t.blend (libraries/lvgl/src/draw/sw/lv_draw_sw.h:32-37): the semantics of lv_draw_sw_blend_basic. */
void lvgpu_blend(lvgpu_lv_draw_ctx_t * draw_ctx, const lvgpu_lv_draw_sw_blend_dsc_t * dsc);
/* lv_draw_ctx_t.wait_for_finish (libraries/lvgl/src/draw/lv_draw.h:129). */
void lvgpu_wait_for_finish(lvgpu_lv_draw_ctx_t * draw_ctx);

/*
 * drv->flush_cb (libraries/lvgl/src/hal/lv_hal_disp.h:107) with the behaviour of lvgl_port_flush_callback
 * (src/lv_port.c:428-572).  drv->user_data must point to a lvgpu_flush_cfg_t, which carries what the reference keeps
 * in lvgl_port_display_ctx_t (src/lv_port.c:431,454): the software rotation and the transport chunk size, plus the
 * sink that stands for esp_lcd_panel_draw_bitmap (x_end/y_end exclusive, as at src/lv_port.c:556).
 * The frame is rotated on the GPU; the chunks handed to draw_bitmap have the same rectangles, order and contents as
 * the reference's.  Ends with lv_disp_flush_ready (src/lv_port.c:571).
 */
typedef struct {
    int32_t sw_rotate;   /* LVGPU_ROT_* */
    uint32_t trans_size; /* pixels per transport chunk; 0 = hand color_p over unrotated in one call */
    int32_t swap_bytes;  /* byte-swap pixels on the way out (LV_COLOR_16_SWAP 0 buffers, big-endian panel) */
    void (*draw_bitmap)(void * user, int x_start, int y_start, int x_end, int y_end, const void * color_data);
    void (*draw_wait)(void * user); /* draw_wait_cb before the first chunk (src/lv_port.c:548-551), may be NULL */
    void * user;
} lvgpu_flush_cfg_t;

void lvgpu_flush_cb(lvgpu_lv_disp_drv_t * drv, const lvgpu_lv_area_t * area, lvgpu_lv_color_t * color_p);

/*
 * LVGPU_ROT_* follow the naming of the reference's flush callback (src/lv_port.c:500-543).  LVGL's OWN software
 * rotation (drv->sw_rotate, libraries/lvgl/src/core/lv_refr.c:1108-1126,1226) names the two quarter turns the other
 * way round: what it flushes for drv->rotated == LV_DISP_ROT_90 is the LVGPU_ROT_270 image and vice versa (checked
 * against whole frames rotated by LVGL itself, tests/golden/lvgl_sw_rotate_swap*.npz).  This returns the LVGPU_ROT_*
 * value that reproduces LVGL's output for a given drv->rotated, or -1 for a value outside 0..3.
 */
int lvgpu_rotation_for_lvgl_sw_rotate(int lv_disp_rot);

/* Counters for tests and integration checks: blend calls executed on the GPU, blend calls handed to the host's
 * lv_draw_sw_blend_basic (requests outside this library's coverage), flush calls, failed calls. */
typedef struct {
    uint64_t blend_gpu, blend_delegated, flush_gpu, errors;
    uint64_t blend_lists;  /* deferred mode: display lists submitted */
    uint64_t list_replays; /* ... of which drawn by replaying the plan of the previous, identical refresh */
    uint64_t submit_ns;    /* host time spent submitting (copies, planning, launch, wait), nanoseconds */
    uint64_t lists_in_place; /* ... of blend_lists: drawn in place in host memory, without a copy in either direction */
} lvgpu_lvgl_stats_t;
void lvgpu_lvgl_get_stats(lvgpu_lvgl_stats_t * out);
void lvgpu_lvgl_reset_stats(void);

#ifdef __cplusplus
}
#endif
#endif /* LVGPU_LVGL_H */
