/*
 * lvgpu.h -- C ABI of liblvgpu.so: LVGL 8.x's software pixel path (fill / alpha blend / RGB565 conversion /
 * flush rotate) as batched HIP kernels for AMD Instinct MI355X (gfx950).
 *
 * Everything here is plain C: pointers, sizes and PODs, no LVGL, HIP or torch types.  Buffers named "device" are
 * HIP device pointers (hipMalloc, lvgpu_malloc, or a torch tensor's data_ptr()); `stream` is a hipStream_t passed
 * as void* (NULL = the default stream).  Every entry point returns LVGPU_OK (0) or a negative LVGPU_ERR_* code and
 * never throws; lvgpu_last_error() gives the text of the last failure on the calling thread.  Calls are
 * asynchronous with respect to the host unless stated otherwise.  Entry points may be called from several host
 * threads as long as each thread uses its own device/stream.  The full-frame kernels look at the stream's CU mask
 * (hipExtStreamGetCUMask): on a stream that owns the whole device they cap how many workgroups share a CU, which
 * raises the HBM rate; a stream confined to some CUs keeps the default.  Results do not depend on it.  The cap is a
 * tuning measured on MI355X under ROCm 7.2; setting LVGPU_NO_RESIDENCY_CAP in the environment (read once, at the first
 * launch) disables it, and tools/check_residency.py compares both settings on the machine at hand.  Two more
 * environment switches exist for measuring, none of them changes a result: LVGPU_NO_BAND (read once) keeps the fused
 * quarter-turn flush of masked fills and maps on the tile kernel instead of the band kernel, LVGPU_ANY_WPB=1..8 (read
 * at every launch) fixes the workgroup size of the any-size quarter-turn kernel (tools/bench_fused_src.py,
 * tools/bench_any_shapes.sh).
 *
 * Pixel vocabulary (libraries/lvgl/src/misc/lv_color.h:208-222): a pixel is lv_color16_t, 2 bytes.
 *   color_16_swap = 1 : LV_COLOR_16_SWAP 1, memory bytes {RRRRRGGG, GGGBBBBB} (big-endian RGB565 = panel wire order)
 *   color_16_swap = 0 : LV_COLOR_16_SWAP 0, little-endian RGB565
 * Areas are LVGL areas: int16 {x1,y1,x2,y2}, inclusive, absolute (libraries/lvgl/src/misc/lv_area.h:43-48).
 *
 * The LVGL-signature plug-in callbacks built on these live in lvgpu_lvgl.h.
 */
#ifndef LVGPU_H
#define LVGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LVGPU_VERSION 0x000100

#define LVGPU_OK 0
#define LVGPU_ERR_ARG (-1)         /* malformed descriptor (NULL buffer, negative count, bad enum, ...) */
#define LVGPU_ERR_HIP (-2)         /* a HIP runtime call failed; see lvgpu_last_error() */
#define LVGPU_ERR_UNSUPPORTED (-3) /* valid LVGL request this library does not implement */
#define LVGPU_ERR_NO_DEVICE (-4)   /* no gfx950 device visible */

/* lv_area_t (libraries/lvgl/src/misc/lv_area.h:43-48), same layout. */
typedef struct {
    int16_t x1, y1, x2, y2;
} lvgpu_area_t;

/* lv_draw_mask_res_t (libraries/lvgl/src/draw/lv_draw_mask.h:35-42). */
enum { LVGPU_MASK_RES_TRANSP = 0, LVGPU_MASK_RES_FULL_COVER = 1, LVGPU_MASK_RES_CHANGED = 2, LVGPU_MASK_RES_UNKNOWN = 3 };

/* lv_blend_mode_t (libraries/lvgl/src/misc/lv_style.h:86-94). */
enum {
    LVGPU_BLEND_NORMAL = 0,
    LVGPU_BLEND_ADDITIVE = 1,
    LVGPU_BLEND_SUBTRACTIVE = 2,
    LVGPU_BLEND_MULTIPLY = 3,
    LVGPU_BLEND_REPLACE = 4
};

/* lv_disp_rot_t as src/lv_port.c:466-546 interprets it (NOT lv_refr.c's own sw_rotate, which mirrors 90/270). */
enum { LVGPU_ROT_NONE = 0, LVGPU_ROT_90 = 1, LVGPU_ROT_180 = 2, LVGPU_ROT_270 = 3 };

/* What `src` of a blend holds. */
enum {
    LVGPU_SRC_FILL = 0,     /* no image: fill with `color` (lv_draw_sw_blend_dsc_t.src_buf == NULL)                  */
    LVGPU_SRC_RGB565 = 1,   /* lv_color_t image, 2 B/px, same byte order as the destination                         */
    LVGPU_SRC_ARGB8888 = 2, /* lv_color32_t image, bytes {B,G,R,A} (libraries/lvgl/src/misc/lv_color.h:224-232):  */
                            /* colour truncated to 565 (lv_color_make, :148-154), A used as the per-pixel mask,     */
                            /* i.e. lv_png.c:244-253 + lv_draw_sw_img.c:249-275 fused in front of map_normal        */
    LVGPU_SRC_RGBA8888 = 3, /* same, bytes {R,G,B,A} as lodepng delivers them                                       */
    LVGPU_SRC_ARGB8565 = 4  /* LV_IMG_CF_TRUE_COLOR_ALPHA at depth 16: 3 B/px {c.lo, c.hi, A}                      */
                            /* (libraries/lvgl/src/draw/sw/lv_draw_sw_img.c:249-275)                                */
};

/* ------------------------------------------------------------------------------------------------------------ */
/* Library / device                                                                                             */
/* ------------------------------------------------------------------------------------------------------------ */

/* Select and initialise a device (hipSetDevice + a first kernel load).  Returns LVGPU_ERR_NO_DEVICE when no HIP
 * device exists.  Optional: every entry point works on the calling thread's current device. */
int lvgpu_init(int device);
int lvgpu_version(void);
const char * lvgpu_last_error(void);
int lvgpu_device_count(void);

/* Device memory helpers so that a C host needs no HIP headers. */
int lvgpu_malloc(void ** dev_ptr, size_t bytes);
int lvgpu_free(void * dev_ptr);
int lvgpu_memcpy_h2d(void * dev_dst, const void * host_src, size_t bytes, void * stream);
int lvgpu_memcpy_d2h(void * host_dst, const void * dev_src, size_t bytes, void * stream);
int lvgpu_memset(void * dev_dst, int value, size_t bytes, void * stream);
int lvgpu_stream_synchronize(void * stream);
/* A stream of the calling thread's current device that does not synchronise with the default stream
 * (hipStreamCreateWithFlags, hipStreamNonBlocking): one per host thread is what a multi-device host uses, see
 * examples/multi_gpu_blend_flush.c.  Released with lvgpu_stream_destroy. */
int lvgpu_stream_create(void ** stream);

/* ------------------------------------------------------------------------------------------------------------ */
/* Blend: lv_draw_sw_blend over a batch of frames                                                               */
/* ------------------------------------------------------------------------------------------------------------ */

/*
 * One lv_draw_sw_blend call (libraries/lvgl/src/draw/sw/lv_draw_sw_blend.c:95-204) applied to n_frames independent
 * destination buffers.  Field-for-field it is lv_draw_ctx_t {buf, buf_area, clip_area}
 * (libraries/lvgl/src/draw/lv_draw.h:59-73) plus lv_draw_sw_blend_dsc_t
 * (libraries/lvgl/src/draw/sw/lv_draw_sw_blend.h:29-39), with pointers replaced by device pointers + per-frame
 * strides.  Strides follow the reference: destination row stride = width(buf_area) (:117), source row stride =
 * width(blend_area) (:140), mask row stride = width(mask_area) (:158).
 *
 * Frame f uses dst + f*dst_frame_stride pixels, src + f*src_frame_stride pixels (0 = one image shared by all
 * frames), mask + f*mask_frame_stride bytes (0 = shared).
 *
 * Memory touched: only the area's pixels are ever written.  The destination is read in aligned 16-byte words, so for an
 * area whose rows do not begin or end on a 16-byte boundary up to 14 bytes beside the row, inside the same aligned
 * 16-byte word, are read and discarded (device allocations are 256-byte granular, so such a word never leaves the
 * allocation the pixels live in).  Source and mask reads stay between the first and the last pixel of the area's
 * rows (stride included).
 *
 * Semantics replaced: lv_draw_sw_blend gate (opa <= LV_OPA_MIN -> no-op; empty clip -> no-op, :95-106),
 * lv_draw_sw_blend_basic mask_res handling (:111-115), fill_normal (:242-375), map_normal (:641-734), and with
 * src_format ARGB8888/RGBA8888/ARGB8565 the colour/alpha split in front of them.  Results are bit-exact.
 */
typedef struct {
    void * dst;               /* device: n_frames destination buffers of lv_color_t                              */
    int64_t dst_frame_stride; /* pixels between consecutive frames' buffers                                      */
    lvgpu_area_t buf_area;    /* draw_ctx->buf_area                                                              */
    lvgpu_area_t clip_area;   /* draw_ctx->clip_area                                                             */
    lvgpu_area_t blend_area;  /* dsc->blend_area                                                                 */
    const void * src;         /* device, or NULL when src_format == LVGPU_SRC_FILL                               */
    int64_t src_frame_stride; /* source pixels between frames, 0 = shared                                        */
    const uint8_t * mask;     /* device A8 mask or NULL (dsc->mask_buf); must be NULL for the *8888/8565 formats */
    int64_t mask_frame_stride; /* bytes between frames, 0 = shared                                               */
    lvgpu_area_t mask_area;   /* dsc->mask_area (ignored when mask == NULL)                                      */
    int32_t n_frames;
    int32_t src_format;       /* LVGPU_SRC_*                                                                     */
    uint16_t color;           /* dsc->color.full (fill colour)                                                   */
    uint8_t mask_res;         /* dsc->mask_res                                                                   */
    uint8_t opa;              /* dsc->opa                                                                        */
    uint8_t blend_mode;       /* dsc->blend_mode (LVGPU_BLEND_*)                                                 */
    uint8_t color_16_swap;    /* LV_COLOR_16_SWAP of the buffers                                                 */
    uint8_t dst_argb8565;     /* 1: dst is the 3-byte ARGB8565 layout {c.lo, c.hi, A} that lv_draw_sw_blend_basic draws   */
                              /* into when disp->driver->screen_transp is set (lv_draw_sw_blend.c:129-133, 177-184):    */
                              /* fill_argb / map_argb semantics; dst_frame_stride stays in pixels (3 bytes each)       */
    uint8_t flags;            /* LVGPU_BLEND_FLAG_* bits, 0 = the behaviour described above                          */
} lvgpu_blend_batch_t;

/*
 * lvgpu_blend_batch_t.flags.
 * LVGPU_BLEND_FLAG_NO_AA  the display has anti-aliasing switched off (lv_disp_drv_t.antialiasing == 0): every mask value
 *     -- the bytes of `mask`, or the alpha of a source with its own alpha, whose alpha plane is the mask of the blend
 *     (libraries/lvgl/src/draw/sw/lv_draw_sw_img.c:133-136) -- is rounded to 0 or 255 before it is used,
 *     `m > 128 ? LV_OPA_COVER : LV_OPA_TRANSP` (libraries/lvgl/src/draw/sw/lv_draw_sw_blend.c:150-156).  The reference
 *     rounds the caller's mask buffer in place; here the values are rounded as they are loaded and the buffer is left as
 *     it is (rounding is idempotent, so a caller that uses the buffer as a mask again sees the same pixels either way; a
 *     caller that reads the bytes back as something else -- LVGL's shadow renderer takes opacities from a buffer it has
 *     blended out of, libraries/lvgl/src/draw/sw/lv_draw_sw_rect.c:572,661 -- has to round them itself, which is what
 *     lvgpu_blend of lvgpu_lvgl.h does instead of setting this flag).  Accepted by
 *     lvgpu_blend_batch, lvgpu_blend_batch_noalloc and lvgpu_blend_flush_batch (which then blends and flushes in two
 *     launches when the rotation is a quarter turn: the fused quarter-turn kernels are compiled without the switch);
 *     lvgpu_list_create answers LVGPU_ERR_UNSUPPORTED for an item that carries it.  Without a mask (or with mask_res
 *     FULL_COVER) it does nothing, as in the reference.
 * LVGPU_BLEND_FLAG_DISCARD_DST  lvgpu_blend_flush_batch only (LVGPU_ERR_ARG anywhere else): the caller does not need the
 *     blended draw buffer after the flush, only the panel image.  That is the reference's own configuration: its port
 *     forces full_refresh (src/lv_port.c:241), so LVGL redraws the whole buffer on every refresh
 *     (libraries/lvgl/src/core/lv_refr.c:633-642) and what the previous refresh left in it is never read.  Where the
 *     blend and the flush run as one pass over memory the write-back of dst is skipped (8 instead of 10 bytes per pixel
 *     for an ARGB8888 image) and dst keeps the pixels it had before the call; where they run as two passes dst holds
 *     the blended pixels as without the flag.  Either way the panel image is the same as without the flag, and the
 *     contents of dst afterwards are unspecified (one of the two).  Not for displays that keep drawing into the buffer:
 *     partial refresh (full_refresh 0 with a draw buffer smaller than the screen still redraws each part whole, which
 *     is fine; direct_mode, where the buffer IS the frame that accumulates, is not) and anything that reads dst back.
 */
#define LVGPU_BLEND_FLAG_NO_AA 0x01
#define LVGPU_BLEND_FLAG_DISCARD_DST 0x04

int lvgpu_blend_batch(const lvgpu_blend_batch_t * dsc, void * stream);

/*
 * lvgpu_blend_batch for callers that cannot allocate device memory: code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc.  It has no counterpart in the reference.  Same arguments, same results; the one
 * piece of device memory lvgpu_blend_batch allocates per call -- one 32-bit scan index per frame, needed by the
 * translucent unmasked fill (lv_draw_sw_blend.c:261-287) to reproduce the reference's first-pixel cache -- lives in a
 * fixed array of LVGPU_NOALLOC_MAX_FRAMES entries (256 KB) that is part of the library's device image, so every GPU
 * the library runs on holds one until the process ends.  The call performs no hipMalloc / hipFree of any kind.
 * n_frames > LVGPU_NOALLOC_MAX_FRAMES is LVGPU_ERR_ARG whatever the descriptor (split the batch).  All calls on one
 * GPU share that GPU's array: calls that may run at the same time on one GPU (different streams, different host
 * threads) race on it; calls on one stream, or otherwise ordered, are fine.
 */
#define LVGPU_NOALLOC_MAX_FRAMES 65536
int lvgpu_blend_batch_noalloc(const lvgpu_blend_batch_t * dsc, void * stream);

/* ------------------------------------------------------------------------------------------------------------ */
/* Flush: the copy / rotate of lvgl_port_flush_callback over a batch of frames                                  */
/* ------------------------------------------------------------------------------------------------------------ */

/*
 * The pixel movement of lvgl_port_flush_callback (src/lv_port.c:428-572) for a flushed area of width x height
 * LVGL pixels, with all its transport chunks concatenated in emission order (the chunks of one frame tile the
 * panel image exactly, so the concatenation IS the panel image):
 *   ROT_NONE: out[y*W + x]               = in[y*W + x]      (:533-538)
 *   ROT_90  : out[x*H + (H-1-y)]         = in[y*W + x]      (:500-505)   panel image is H wide, W tall
 *   ROT_180 : out[(H-1-y)*W + (W-1-x)]   = in[y*W + x]      (:522-527)
 *   ROT_270 : out[(W-1-x)*H + y]         = in[y*W + x]      (:511-516)
 * swap_bytes != 0 additionally byte-swaps every pixel on the way out: that is the conversion of a
 * LV_COLOR_16_SWAP 0 draw buffer into the panel's big-endian wire order (src/display.h:26-30); with
 * LV_COLOR_16_SWAP 1 buffers the reference has no swap pass and swap_bytes must be 0.
 * in and out must not overlap.
 */
typedef struct {
    const void * in;          /* device: n_frames rendered buffers, width*height lv_color_t each                 */
    int64_t in_frame_stride;  /* pixels                                                                          */
    void * out;               /* device: n_frames panel images, width*height pixels each                         */
    int64_t out_frame_stride; /* pixels                                                                          */
    int32_t width, height;    /* of the flushed area (LVGL logical orientation)                                  */
    int32_t rotation;         /* LVGPU_ROT_*                                                                     */
    int32_t swap_bytes;
    int32_t n_frames;
    int32_t reserved;
} lvgpu_flush_batch_t;

int lvgpu_flush_rotate_batch(const lvgpu_flush_batch_t * dsc, void * stream);

/*
 * Blend followed by the flush of the whole destination buffer (buf_area), in one pass over memory where the
 * shapes allow it: the blended pixels are written back to dst (LVGL's draw buffer is mutated in place) AND written
 * rotated to out.  Result == lvgpu_blend_batch(blend) then lvgpu_flush_rotate_batch({dst -> out, W = width(buf_area),
 * H = height(buf_area)}).
 */
typedef struct {
    lvgpu_blend_batch_t blend;
    void * out;               /* device: n_frames panel images                                                   */
    int64_t out_frame_stride; /* pixels                                                                          */
    int32_t rotation;         /* LVGPU_ROT_*                                                                     */
    int32_t swap_bytes;
} lvgpu_blend_flush_batch_t;

int lvgpu_blend_flush_batch(const lvgpu_blend_flush_batch_t * dsc, void * stream);

/*
 * Display list: n_items blend descriptors applied in order to the same batch of draw buffers, as one launch.
 * This is the batched form of the call sequence LVGL's producers issue for a scene -- one lv_draw_sw_blend per
 * object part or per row (libraries/lvgl/src/draw/sw/lv_draw_sw_rect.c:118-330 draw_bg, lv_draw_sw_letter.c:246-398
 * draw_letter_normal, lv_draw_sw_line.c, lv_draw_sw_arc.c, lv_draw_sw_img.c:137-195) -- recorded once and replayed
 * over thousands of buffers.  Result == lvgpu_blend_batch(&items[0]), lvgpu_blend_batch(&items[1]), ... in order.
 * Every item must name the same dst, dst_frame_stride, buf_area, n_frames, color_16_swap and dst_argb8565; clip
 * areas, sources, masks, colours, opacities and blend modes are per item.  src/mask buffers must not overlap dst.
 * The draw buffers are read once and written once per list (tiles of the buffer stay on chip while the tile's items
 * are applied), except that an item which needs a whole-area pre-pass (a translucent unmasked fill whose result on
 * black depends on scan order, lv_draw_sw_blend.c:262-283) or a shape the tile kernels do not take (width not a
 * multiple of 8 pixels -- 16 for an ARGB8565 destination --, unaligned dst) is run through the lvgpu_blend_batch
 * path in its place in the order.  items is host memory and may be released when the call returns.  The tables the
 * kernel walks are planned on the host at every call (about 0.2 ms for 1600 items; see lvgpu_list_create for lists
 * that repeat) and travel through a pinned staging buffer that each calling thread keeps and releases when it exits.
 */
int lvgpu_blend_list_batch(const lvgpu_blend_batch_t * items, int32_t n_items, void * stream);

/* The same followed by the flush of the whole buffer (as lvgpu_blend_flush_batch): panel images go to out.  The
 * geometry (dst, buf_area, n_frames ...) is taken from geom, whose other fields are ignored; n_items may be 0. */
int lvgpu_blend_list_flush_batch(const lvgpu_blend_batch_t * geom, const lvgpu_blend_batch_t * items, int32_t n_items,
                                 void * out, int64_t out_frame_stride, int32_t rotation, int32_t swap_bytes, void * stream);

/*
 * What a display list would cost, without drawing anything (host arithmetic only; no device memory is touched, the
 * pointers in the descriptors are only compared and checked for alignment).  geom may be NULL (= items[0]); flush != 0
 * describes lvgpu_blend_list_flush_batch with `rotation`, else lvgpu_blend_list_batch.
 *   tiles / tiles_selected / tiles_overwritten : 64x64 tiles of the buffer; visited; never read because an opaque item
 *                                                covers them
 *   items_listed / items_standalone / items_noop: drawn by the tile pass; run through the single-descriptor path in
 *                                                their place; dropped by the gates of lv_draw_sw_blend (opa, clip, mask_res)
 *   launches                                   : kernel launches in total
 *   item_steps / batch_steps                   : steps of the tile lists (one item over a large region / one batch of chunks)
 *   chunks_full / chunks_partial               : 8-pixel pieces in the batches
 *   px_item_steps + px_chunks + px_culled      : == the clipped pixels of all listed items; px_culled is what per-tile
 *                                                occlusion culling removed
 */
typedef struct {
    int32_t tiles, tiles_selected, tiles_overwritten;
    int32_t items_listed, items_standalone, items_noop;
    int32_t launches, flush_fused;
    int64_t item_steps, batch_steps, chunks_full, chunks_partial;
    int64_t px_item_steps, px_chunks, px_culled;
} lvgpu_list_plan_t;

int lvgpu_blend_list_plan(const lvgpu_blend_batch_t * geom, const lvgpu_blend_batch_t * items, int32_t n_items, int32_t flush,
                          int32_t rotation, lvgpu_list_plan_t * plan);

/*
 * A display list planned once and replayed many times.  lvgpu_list_create does everything lvgpu_blend_list_batch /
 * lvgpu_blend_list_flush_batch do up to the launch (argument checks, tiling, batching, culling) and leaves the
 * tables on the device; lvgpu_list_replay is then a single kernel launch that allocates and copies nothing --
 * for scenes that are drawn over and over (every refresh of a static screen, the same overlay on batch after
 * batch).  The descriptors' src and mask buffers are referenced, not copied: they must stay where they are.
 * Only lists that the tile pass takes whole can be compiled (see lvgpu_blend_list_batch); otherwise
 * LVGPU_ERR_UNSUPPORTED, and the uncompiled entry points remain.  flush != 0: replay also writes the panel images.
 * lvgpu_list_replay: dst == NULL / n_frames == 0 keep what geom named; another dst must have the same geometry
 * (buf_area, frame stride) and 16-byte alignment.  More frames than were recorded are accepted only when every source
 * and mask of the list is shared by all frames (frame stride 0); otherwise LVGPU_ERR_ARG, since frame n_frames - 1
 * would read past the recorded buffers.  out / out_frame_stride are used when the list was created with flush.
 * A list belongs to the device that was current when it was created; replaying it with another device current is
 * LVGPU_ERR_ARG.
 * There is no counterpart in the reference, whose producers re-issue their descriptors on every refresh.
 */
typedef struct lvgpu_list lvgpu_list_t;
int lvgpu_list_create(const lvgpu_blend_batch_t * geom, const lvgpu_blend_batch_t * items, int32_t n_items, int32_t flush,
                      int32_t rotation, int32_t swap_bytes, lvgpu_list_t ** list);
int lvgpu_list_replay(const lvgpu_list_t * list, void * dst, int32_t n_frames, void * out, int64_t out_frame_stride,
                      void * stream);
int lvgpu_list_destroy(lvgpu_list_t * list);

/*
 * The transport chunks of the reference's flush callback (src/lv_port.c:466-566): the flushed area is cut into pieces of
 * at most trans_size pixels -- whole source columns for 90/270 degrees (:482-487, :500-521), whole rows otherwise
 * (:488-494, :522-543) -- and each piece is sent to the panel as one rectangle.  Every piece is a contiguous slice of
 * the panel image that lvgpu_flush_rotate_batch / lvgpu_blend_flush_batch write for the area, so a transport can send
 * straight from that image: out[i] gets the rectangle (end exclusive, as esp_lcd_panel_draw_bitmap takes it, :556) and
 * the slice.  hor_res / ver_res are drv->hor_res / drv->ver_res.  Host arithmetic only.  Returns the number of
 * chunks (at most max_chunks are written; pass out == NULL to count) or a negative LVGPU_ERR_*.
 */
typedef struct {
    int32_t x_start, y_start, x_end, y_end; /* panel rectangle, end exclusive                                          */
    int64_t offset_px;                      /* first pixel of the piece in the area's panel image                      */
    int32_t n_px;
    int32_t reserved;
} lvgpu_flush_chunk_t;

int lvgpu_flush_chunks(const lvgpu_area_t * area, int32_t hor_res, int32_t ver_res, int32_t rotation, uint32_t trans_size,
                       lvgpu_flush_chunk_t * out, int32_t max_chunks);

/*
 * One transport chunk as the reference's AXS15231B panel driver puts it on the bus (panel_axs15231b_draw_bitmap,
 * src/esp_lcd_axs15231b.c:292-337): the gaps are added to the rectangle (:301-304); CASET carries the column window as
 * two big-endian 16-bit values, last column inclusive (:307-314); the plain-SPI interface also sends RASET with the row
 * window (:316-325), the QSPI interface does not; the pixels follow under RAMWR when the chunk starts at panel row 0
 * and under RAMWRC ("write memory continue") otherwise (:330-334), x_end - x_start times y_end - y_start times
 * bits_per_pixel / 8 bytes of them (:328-329), unchanged (big-endian RGB565 = what an LV_COLOR_16_SWAP 1 draw buffer
 * or a swap_bytes flush holds).  cmd is the value the driver hands to esp_lcd_panel_io_tx_param / _tx_color: the
 * command itself, or for QSPI (opcode << 24) | (command << 8) with opcode 0x02 for parameters and 0x32 for pixels
 * (tx_param / tx_color, :173-193).  out receives 2 (QSPI) or 3 transactions; the last one is the pixel transfer, whose
 * data is chunk->n_px pixels starting at chunk->offset_px of the panel image.  Host arithmetic only; a recorder or a
 * transport on the host replays them in order.  Returns the number of transactions or a negative LVGPU_ERR_*.
 */
enum {
    LVGPU_LCD_CMD_CASET = 0x2A,
    LVGPU_LCD_CMD_RASET = 0x2B,
    LVGPU_LCD_CMD_RAMWR = 0x2C,
    LVGPU_LCD_CMD_RAMWRC = 0x3C
};
typedef struct {
    uint32_t cmd;        /* as handed to the panel IO layer (see above)                                            */
    uint32_t param_len;  /* 4 for CASET / RASET, 0 for the pixel transfer                                          */
    uint8_t param[4];
    int32_t is_color;    /* 1: the pixel transfer                                                                  */
    int64_t data_offset_px;
    int64_t data_len;    /* bytes                                                                                  */
} lvgpu_panel_txn_t;

int lvgpu_axs15231b_transactions(const lvgpu_flush_chunk_t * chunk, int32_t x_gap, int32_t y_gap, int32_t qspi,
                                 int32_t bits_per_pixel, lvgpu_panel_txn_t out[3]);

/* ------------------------------------------------------------------------------------------------------------ */
/* Colour conversion (libraries/lvgl/src/misc/lv_color.h)                                                       */
/* ------------------------------------------------------------------------------------------------------------ */

/* lv_color_hex (:599-631) / lv_color_make (:594-597) over n pixels: 0x00RRGGBB words -> lv_color16_t. */
int lvgpu_color_from_rgb888(const uint32_t * dev_xrgb, uint16_t * dev_out, size_t n, int color_16_swap, void * stream);
/* lv_color_to32 (:374-427) over n pixels: lv_color16_t -> 0xFFRRGGBB words. */
int lvgpu_color_to32(const uint16_t * dev_in, uint32_t * dev_out, size_t n, int color_16_swap, void * stream);
/* The ARGB8888 -> (lv_color16_t, A8) split of lv_png.c:244-253 + lv_draw_sw_img.c:249-275 as a stand-alone pass.
 * src_format is LVGPU_SRC_ARGB8888, LVGPU_SRC_RGBA8888 or LVGPU_SRC_ARGB8565. */
int lvgpu_image_split(const void * dev_src, int src_format, uint16_t * dev_color, uint8_t * dev_alpha, size_t n,
                      int color_16_swap, void * stream);
/*
 * What lv_draw_sw_img_decoded puts into its colour and alpha buffers before it blends them, for a whole image:
 * convert_cb (libraries/lvgl/src/draw/sw/lv_draw_sw_img.c:203-298) followed by the recolor pass (:163-174).
 * src_format LVGPU_SRC_RGB565 is LV_IMG_CF_TRUE_COLOR (alpha 0xFF everywhere, :215-216) or, with
 * opts->chroma_keyed, LV_IMG_CF_TRUE_COLOR_CHROMA_KEYED (alpha 0 where the pixel equals opts->chroma_key, :230-247);
 * the formats with their own alpha are split as in lvgpu_image_split.  With opts->recolor_opa > LV_OPA_MIN every
 * colour becomes lv_color_mix_premult(lv_color_premult(recolor, recolor_opa), colour, 255 - recolor_opa).
 * The two planes are then drawn with lvgpu_blend_batch (src = colour plane, LVGPU_SRC_RGB565; mask = alpha plane;
 * mask_res FULL_COVER for a plain LV_IMG_CF_TRUE_COLOR image, :128-131).  Image assets are converted once and the
 * planes shared by all frames (src_frame_stride = mask_frame_stride = 0).  opts == NULL: no key, no recolor.
 */
typedef struct {
    int32_t chroma_keyed;
    uint16_t chroma_key;  /* LV_COLOR_CHROMA_KEY as stored (lv_color_t.full) */
    uint16_t recolor;     /* lv_draw_img_dsc_t.recolor.full                  */
    uint8_t recolor_opa;  /* lv_draw_img_dsc_t.recolor_opa                   */
    uint8_t reserved[3];
} lvgpu_image_opts_t;

int lvgpu_image_convert(const void * dev_src, int src_format, uint16_t * dev_color, uint8_t * dev_alpha, size_t n,
                        int color_16_swap, const lvgpu_image_opts_t * opts, void * stream);
/* lv_color_fill (libraries/lvgl/src/misc/lv_color.c:36-77). */
int lvgpu_color_fill(uint16_t * dev_buf, uint16_t color, size_t n, void * stream);
/* lv_draw_sw_buffer_copy (libraries/lvgl/src/draw/sw/lv_draw_sw.c:81-104) over n_frames buffers. */
int lvgpu_buffer_copy_batch(void * dev_dst, int64_t dst_frame_stride, int32_t dst_stride, lvgpu_area_t dst_area,
                            const void * dev_src, int64_t src_frame_stride, int32_t src_stride, lvgpu_area_t src_area,
                            int32_t n_frames, void * stream);

/* ------------------------------------------------------------------------------------------------------------ */
/* Synthetic inputs and checksums (measurement support)                                                         */
/* ------------------------------------------------------------------------------------------------------------ */

enum {
    LVGPU_SYNTH_U16 = 0,      /* uniform lv_color16_t                                                            */
    LVGPU_SYNTH_MASK_UI = 1,  /* A8: P(0)=1/3, P(255)=1/3, else uniform 1..254                                   */
    LVGPU_SYNTH_MASK_UNI = 2, /* A8: uniform 0..255                                                              */
    LVGPU_SYNTH_ARGB_UI = 3,  /* u32: uniform RGB, A from the UI-like mask distribution                          */
    LVGPU_SYNTH_ARGB_UNI = 4  /* u32: uniform                                                                    */
};
/* Element i (global index first_index + i) is a pure function of (kind, seed, global index): counter-based
 * splitmix64 (out[i] = f(seed + (first_index+i+1) * 0x9E3779B97F4A7C15)), so shards are reproducible for any frame
 * partition. */
int lvgpu_synth(void * dev_out, int kind, uint64_t seed, uint64_t first_index, size_t count, void * stream);
/* Order-independent 64-bit checksum of n 16-bit (elem_bytes=2), 8-bit (1) or 32-bit (4) elements:
 * sum over i of splitmix64(first_index + i, value_i) mod 2^64.  *dev_sum64 is ADDED to (zero it first). */
int lvgpu_checksum(const void * dev_in, int elem_bytes, uint64_t first_index, size_t count, uint64_t * dev_sum64,
                   void * stream);

/* A stream restricted to the compute units whose bits are set in mask_words (hipExtStreamCreateWithCUMask): used
 * to measure the path on a fraction of the GPU, e.g. the 32 CUs of one XCD ("1/8 GPU"). */
int lvgpu_stream_create_cu_mask(void ** stream, const uint32_t * mask_words, int n_words);
int lvgpu_stream_destroy(void * stream);
/* Launches n_blocks workgroups on `stream` and adds, per XCD id (0..7), the number of workgroups that ran there to
 * dev_counts8[0..7] (device memory, zero it first): tells which XCDs a CU mask actually selects. */
int lvgpu_xcc_census(uint32_t * dev_counts8, int n_blocks, void * stream);

/* Counters since the library was loaded, for tests that pin properties no result shows: the device and pinned-memory
 * allocations the library's entry points have made (hipMalloc, hipMallocAsync, hipHostMalloc; lvgpu_malloc included,
 * the LVGL callbacks of lvgpu_lvgl.h not) -- lvgpu_blend_batch_noalloc must leave it unchanged -- and the launches of
 * the scan-order pre-pass of the translucent unmasked fill (the one user of that entry point's fixed scratch array). */
typedef struct {
    uint64_t alloc_calls;
    uint64_t first_nonblack_launches;
    uint64_t band_pad_bytes; /* dynamic LDS the calling thread's last band-kernel launch asked for on top of its 64 KB  */
                             /* (a residency tuning that is dropped where the device would refuse the sum;           */
                             /* LVGPU_MAX_LDS_PER_BLOCK in the environment stands in for the device's limit)          */
} lvgpu_debug_counters_t;
void lvgpu_debug_counters(lvgpu_debug_counters_t * out);

/* Name of the kernel the last launch on this thread used (diagnostics / tests that assert the fast path ran). */
const char * lvgpu_last_kernel(void);

#ifdef __cplusplus
}
#endif

#endif /* LVGPU_H */
