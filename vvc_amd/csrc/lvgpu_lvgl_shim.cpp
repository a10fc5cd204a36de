// lvgpu_lvgl_shim.cpp -- LVGL-signature callbacks (include/lvgpu_lvgl.h) on top of the batched C ABI.
//
// Single-frame, host-buffer versions of the plug-in points of LVGL 8.x: lv_draw_sw_ctx_t.blend,
// lv_draw_ctx_t.wait_for_finish, lv_disp_drv_t.draw_ctx_init/deinit/size and lv_disp_drv_t.flush_cb.
// Immediate mode: each blend call stages the touched rectangle into device memory, runs the batch kernels with
// n_frames = 1 and copies the result back before it returns.
// Deferred mode (lvgpu_lvgl_set_deferred; the default for displays flushed by lvgpu_flush_cb): blend calls only copy
// their source/mask rectangles into a queue; the
// queue is submitted as ONE display list (draw buffer up, one launch, draw buffer down) when something is about
// to look at the buffer -- lvgpu_flush_cb, lvgpu_wait_for_finish, the layer and buffer_copy hooks, a change of
// draw_ctx->buf, context deinit.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/lvgpu_lvgl.h"

namespace {

typedef void (*init_ctx_fn)(lvgpu_lv_disp_drv_t *, lvgpu_lv_draw_ctx_t *);
typedef void (*blend_fn)(lvgpu_lv_draw_ctx_t *, const lvgpu_lv_draw_sw_blend_dsc_t *);
typedef void (*flush_ready_fn)(lvgpu_lv_disp_drv_t *);

init_ctx_fn g_sw_init_ctx = nullptr;
blend_fn g_sw_blend_basic = nullptr;
flush_ready_fn g_flush_ready = nullptr;
int g_swap = 1;
int g_deferred = -1;  // 1 on, 0 off, -1: on for displays flushed by lvgpu_flush_cb
int g_inject_fail = 0;  // > 0: the n-th immediate blend from now fails before it touches the GPU (lvgpu_lvgl_inject_failure)
std::atomic<uint64_t> g_blend_gpu{0}, g_blend_delegated{0}, g_flush_gpu{0}, g_errors{0}, g_lists{0}, g_replays{0}, g_submit_ns{0}, g_in_place{0};

typedef void * (*layer_init_fn)(lvgpu_lv_draw_ctx_t *, void *, int);
typedef void (*layer_adjust_fn)(lvgpu_lv_draw_ctx_t *, void *, int);
typedef void (*layer_blend_fn)(lvgpu_lv_draw_ctx_t *, void *, const void *);
typedef void (*layer_destroy_fn)(lvgpu_lv_draw_ctx_t *, void *);
typedef void (*buffer_copy_fn)(lvgpu_lv_draw_ctx_t *, void *, int16_t, const lvgpu_lv_area_t *, void *, int16_t,
                               const lvgpu_lv_area_t *);

// One queued blend: the clipped rectangle and where its private copies of source and mask lie in the arena.
// Compared bytewise between refreshes (every byte is initialised, padding included: see make_item).
struct QItem {
    lvgpu_area_t area;
    uint64_t src_off, mask_off;
    uint16_t color;
    uint8_t opa, mode, has_src, has_mask;
    uint8_t flags;  // LVGPU_BLEND_FLAG_* of the item
    uint8_t pad[1];
};

// Growable pinned host memory: the sources and masks of the queued blends are copied straight into it, so the one
// transfer to the device is a DMA from pinned memory.
struct PinnedArena {
    uint8_t * data = nullptr;
    size_t size = 0, cap = 0;
    uint8_t * grow_to(size_t new_size);  // keeps the contents; returns data
    void release()
    {
        if(data) (void)hipHostFree(data);
        data = nullptr;
        size = cap = 0;
    }
};

struct Queue {
    std::vector<QItem> items;
    PinnedArena arena;
    std::vector<lvgpu_blend_batch_t> batch;
    void * buf = nullptr;          // the host draw buffer the queued items draw into
    lvgpu_area_t buf_area{0, 0, 0, 0};
    bool transp = false;
    bool covered = false;          // some queued item overwrites the whole buffer: its previous contents do not matter
    void * d_fb = nullptr;
    size_t fb_cap = 0;
    void * d_arena = nullptr;
    size_t arena_cap = 0;
    // A refresh that queues exactly what the previous one queued (same rectangles, opacities, colours, arena layout --
    // the usual case for a UI between two changes of its layout; the bytes of the masks and images may differ) is drawn
    // by replaying the list planned for the previous one.
    std::vector<QItem> prev_items;
    const void * prev_arena = nullptr;  // where the previous submit's launch read its sources and masks (device address)
    lvgpu_area_t prev_area{0, 0, 0, 0};
    bool prev_transp = false;
    int prev_sig = -1;             // -1: nothing recorded; else 0 = no flush, 1 + rot * 2 + swap = fused flush
    int prev_seen = 0;             // consecutive submits with this content
    lvgpu_list_t * compiled = nullptr;
    bool uncompilable = false;
    // the software implementations behind the hooks this library wraps
    layer_init_fn sw_layer_init = nullptr;
    layer_adjust_fn sw_layer_adjust = nullptr;
    layer_blend_fn sw_layer_blend = nullptr;
    layer_destroy_fn sw_layer_destroy = nullptr;
    buffer_copy_fn sw_buffer_copy = nullptr;
};

// lv_draw_sw_ctx_t (libraries/lvgl/src/draw/sw/lv_draw_sw.h:32-37) extended the way GPU back-ends extend it
// (libraries/lvgl/src/draw/stm32_dma2d/lv_gpu_stm32_dma2d.h: typedef lv_draw_sw_ctx_t lv_draw_stm32_dma2d_ctx_t).
struct ShimCtx {
    lvgpu_lv_draw_ctx_t base_draw;
    blend_fn blend;
    // --- extension ---
    uint32_t magic;
    lvgpu_lv_disp_drv_t * drv;
    hipStream_t stream;
    void * d_dst;
    void * d_src;
    void * d_mask;
    size_t cap_px;
    Queue * q;  // deferred mode only
};
constexpr uint32_t MAGIC = 0x4C564750u;  // "LVGP"

[[noreturn]] void die(const char * what, hipError_t e)
{
    // The callbacks return void; where the call cannot be handed to the host's own renderer any more (deferred mode,
    // a result already on its way back) a failure must not go unnoticed.
    fprintf(stderr, "lvgpu: %s failed: %s -- aborting\n", what, hipGetErrorString(e));
    g_errors++;
    abort();
}
#define HIP_OR_DIE(expr)                     \
    do {                                     \
        hipError_t e_ = (expr);              \
        if(e_ != hipSuccess) die(#expr, e_); \
    } while(0)

void * lookup(const char * name) { return dlsym(RTLD_DEFAULT, name); }

uint8_t * PinnedArena::grow_to(size_t new_size)
{
    if(new_size > cap) {
        size_t want = cap ? cap : (size_t)1 << 20;
        while(want < new_size) want *= 2;
        uint8_t * p = nullptr;
        HIP_OR_DIE(hipHostMalloc(reinterpret_cast<void **>(&p), want, hipHostMallocDefault));
        if(size) memcpy(p, data, size);
        if(data) HIP_OR_DIE(hipHostFree(data));
        data = p;
        cap = want;
    }
    size = new_size;
    return data;
}

// The address under which the device sees a piece of pinned host memory (lvgpu_lvgl_buf_alloc, the queue's arena, the
// flush staging); nullptr for ordinary memory.
void * device_view(void * host)
{
    void * d = nullptr;
    if(host == nullptr || hipHostGetDevicePointer(&d, host, 0) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return d;
}

void drop_compiled(Queue * q)
{
    if(q->compiled) (void)lvgpu_list_destroy(q->compiled);
    q->compiled = nullptr;
    q->uncompilable = false;
    q->prev_sig = -1;
    q->prev_seen = 0;
}

inline int aw(const lvgpu_lv_area_t * a) { return a->x2 - a->x1 + 1; }
inline int ah(const lvgpu_lv_area_t * a) { return a->y2 - a->y1 + 1; }

void flush_ready(lvgpu_lv_disp_drv_t * drv)
{
    if(g_flush_ready == nullptr) g_flush_ready = (flush_ready_fn)lookup("lv_disp_flush_ready");
    if(g_flush_ready) g_flush_ready(drv);
    else if(drv->draw_buf) {  // libraries/lvgl/src/hal/lv_hal_disp.c:519-522
        drv->draw_buf->flushing = 0;
        drv->draw_buf->flushing_last = 0;
    }
}

// What lvgpu_flush_cb wants from a submit that happens on its behalf: the panel image of the whole draw buffer.
struct FlushReq {
    void * d_out;      // device staging for the panel image
    uint16_t * panel;  // pinned host memory that receives it
    int rot, swap_bytes;
};

// Submit the queue: the whole list in one launch.  With fr the same launch also writes the rotated panel image (the
// tiles are flushed from on-chip memory).
// In place (a draw buffer in pinned memory, i.e. one that came from lvgpu_lvgl_buf_alloc): the launch reads the
// sources and masks from the pinned arena, reads the draw buffer where no queued item overwrites it, and writes the
// buffer and the panel image, all in host memory over the bus -- no copy is issued, the submission is one launch and
// one wait.
// Staged (any other buffer: LVGL's own layer buffers, a display buffer from malloc; or LVGPU_LVGL_STAGED in the
// environment): arena and, unless a queued item overwrites all of it, draw buffer to the device, the launch, draw
// buffer and panel image back.
void submit(ShimCtx * s, const FlushReq * fr = nullptr)
{
    Queue * q = s->q;
    if(q == nullptr || q->items.empty()) return;
    const auto t_start = std::chrono::steady_clock::now();
    const int W = q->buf_area.x2 - q->buf_area.x1 + 1, H = q->buf_area.y2 - q->buf_area.y1 + 1;
    const size_t bpp = q->transp ? 3 : 2;
    const size_t fb_bytes = (size_t)W * H * bpp;
    hipStream_t st = s->stream;
    void * fb = nullptr;     // where the launch draws ...
    void * arena = nullptr;  // ... where it finds sources and masks ...
    void * out = nullptr;    // ... and where the panel image goes
    bool in_place = false;
    if(getenv("LVGPU_LVGL_STAGED") == nullptr) {
        fb = device_view(q->buf);
        arena = q->arena.size ? device_view(q->arena.data) : nullptr;
        out = fr ? device_view(fr->panel) : nullptr;
        in_place = fb && (arena || q->arena.size == 0) && (!fr || out);
    }
    if(!in_place) {
        if(fb_bytes > q->fb_cap) {
            drop_compiled(q);
            if(q->d_fb) HIP_OR_DIE(hipFree(q->d_fb));
            HIP_OR_DIE(hipMalloc(&q->d_fb, fb_bytes));
            q->fb_cap = fb_bytes;
        }
        if(q->arena.size > q->arena_cap) {
            drop_compiled(q);
            if(q->d_arena) HIP_OR_DIE(hipFree(q->d_arena));
            q->arena_cap = q->arena.size * 2;
            HIP_OR_DIE(hipMalloc(&q->d_arena, q->arena_cap));
        }
        fb = q->d_fb;
        arena = q->d_arena;
        out = fr ? fr->d_out : nullptr;
        if(!q->covered) HIP_OR_DIE(hipMemcpyAsync(q->d_fb, q->buf, fb_bytes, hipMemcpyHostToDevice, st));
        if(q->arena.size) HIP_OR_DIE(hipMemcpyAsync(q->d_arena, q->arena.data, q->arena.size, hipMemcpyHostToDevice, st));
    }

    // the same content as the previous submit?
    const int sig = fr ? 1 + fr->rot * 2 + (fr->swap_bytes ? 1 : 0) : 0;
    // (the draw buffer is not part of the key: a display with two draw buffers alternates between them from refresh to
    // refresh, and a compiled list is replayed over any buffer of the same geometry)
    const bool same = q->prev_sig == sig && q->prev_arena == arena && q->prev_transp == q->transp &&
                      memcmp(&q->prev_area, &q->buf_area, sizeof(lvgpu_area_t)) == 0 && q->prev_items.size() == q->items.size() &&
                      memcmp(q->prev_items.data(), q->items.data(), q->items.size() * sizeof(QItem)) == 0;
    if(!same) {
        if(q->compiled) (void)lvgpu_list_destroy(q->compiled);
        q->compiled = nullptr;
        q->uncompilable = false;
        q->prev_items = q->items;
        q->prev_arena = arena;
        q->prev_area = q->buf_area;
        q->prev_transp = q->transp;
        q->prev_sig = sig;
        q->prev_seen = 1;
    }
    else if(q->prev_seen < 1000000) q->prev_seen++;

    int rc = LVGPU_OK;
    bool drawn = false;
    if(same && q->compiled == nullptr && !q->uncompilable) q->batch.clear();  // marshalled below, then compiled
    if(!(same && q->compiled)) {
        q->batch.resize(q->items.size());
        for(size_t i = 0; i < q->items.size(); i++) {
            const QItem & it = q->items[i];
            lvgpu_blend_batch_t & b = q->batch[i];
            memset(&b, 0, sizeof(b));
            b.dst = fb;
            b.dst_frame_stride = (int64_t)W * H;
            b.buf_area = q->buf_area;
            b.clip_area = it.area;
            b.blend_area = it.area;
            b.mask_area = it.area;
            b.src = it.has_src ? reinterpret_cast<uint8_t *>(arena) + it.src_off : nullptr;
            b.src_format = it.has_src ? LVGPU_SRC_RGB565 : LVGPU_SRC_FILL;
            b.mask = it.has_mask ? reinterpret_cast<uint8_t *>(arena) + it.mask_off : nullptr;
            b.mask_res = LVGPU_MASK_RES_CHANGED;
            b.n_frames = 1;
            b.color = it.color;
            b.opa = it.opa;
            b.blend_mode = it.mode;
            b.color_16_swap = (uint8_t)g_swap;
            b.dst_argb8565 = q->transp ? 1 : 0;
            b.flags = it.flags;
        }
        if(same && !q->uncompilable) {  // second refresh in a row with this content: plan it once and keep the plan
            rc = lvgpu_list_create(&q->batch[0], q->batch.data(), (int32_t)q->batch.size(), fr ? 1 : 0, fr ? fr->rot : 0,
                                   fr ? fr->swap_bytes : 0, &q->compiled);
            if(rc != LVGPU_OK) {  // e.g. a shape the tile pass does not take: the uncompiled call below draws it
                q->compiled = nullptr;
                q->uncompilable = true;
                rc = LVGPU_OK;
            }
        }
    }
    if(q->compiled) {
        rc = lvgpu_list_replay(q->compiled, fb, 0, out, (int64_t)W * H, st);
        drawn = rc == LVGPU_OK;
        if(drawn) g_replays++;
    }
    if(!drawn && rc == LVGPU_OK)
        rc = fr ? lvgpu_blend_list_flush_batch(&q->batch[0], q->batch.data(), (int32_t)q->batch.size(), out, (int64_t)W * H,
                                               fr->rot, fr->swap_bytes, st)
                : lvgpu_blend_list_batch(q->batch.data(), (int32_t)q->batch.size(), st);
    if(rc != LVGPU_OK) {
        fprintf(stderr, "lvgpu: drawing the queued display list failed (%d): %s -- aborting\n", rc, lvgpu_last_error());
        g_errors++;
        abort();
    }
    if(!in_place) {
        HIP_OR_DIE(hipMemcpyAsync(q->buf, q->d_fb, fb_bytes, hipMemcpyDeviceToHost, st));
        if(fr) HIP_OR_DIE(hipMemcpyAsync(fr->panel, fr->d_out, fb_bytes, hipMemcpyDeviceToHost, st));
    }
    else g_in_place++;
    HIP_OR_DIE(hipStreamSynchronize(st));
    q->items.clear();
    q->arena.size = 0;
    q->covered = false;
    g_lists++;
    g_submit_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_start).count();
}

inline ShimCtx * shim_of(lvgpu_lv_draw_ctx_t * c)
{
    ShimCtx * s = reinterpret_cast<ShimCtx *>(c);
    return (s && s->magic == MAGIC) ? s : nullptr;
}

// Hooks at which LVGL's own code reads, replaces or frees the buffer the queue draws into
// (libraries/lvgl/src/draw/sw/lv_draw_sw_layer.c:42-141, libraries/lvgl/src/draw/sw/lv_draw_sw.c:80-101).
void * hook_layer_init(lvgpu_lv_draw_ctx_t * c, void * layer, int flags)
{
    ShimCtx * s = shim_of(c);
    submit(s);
    return s->q->sw_layer_init ? s->q->sw_layer_init(c, layer, flags) : nullptr;
}
void hook_layer_adjust(lvgpu_lv_draw_ctx_t * c, void * layer, int flags)
{
    ShimCtx * s = shim_of(c);
    submit(s);
    if(s->q->sw_layer_adjust) s->q->sw_layer_adjust(c, layer, flags);
}
void hook_layer_blend(lvgpu_lv_draw_ctx_t * c, void * layer, const void * dsc)
{
    ShimCtx * s = shim_of(c);
    submit(s);  // the layer buffer is about to be read as an image
    if(s->q->sw_layer_blend) s->q->sw_layer_blend(c, layer, dsc);
}
void hook_layer_destroy(lvgpu_lv_draw_ctx_t * c, void * layer)
{
    ShimCtx * s = shim_of(c);
    submit(s);
    if(s->q->sw_layer_destroy) s->q->sw_layer_destroy(c, layer);
}
void hook_buffer_copy(lvgpu_lv_draw_ctx_t * c, void * dest_buf, int16_t dest_stride, const lvgpu_lv_area_t * dest_area,
                      void * src_buf, int16_t src_stride, const lvgpu_lv_area_t * src_area)
{
    ShimCtx * s = shim_of(c);
    submit(s);
    if(s->q->sw_buffer_copy) s->q->sw_buffer_copy(c, dest_buf, dest_stride, dest_area, src_buf, src_stride, src_area);
}

}  // namespace

extern "C" {

void lvgpu_lvgl_bind(void * init_ctx, void * blend_basic, void * flush_ready_p)
{
    g_sw_init_ctx = (init_ctx_fn)init_ctx;
    g_sw_blend_basic = (blend_fn)blend_basic;
    g_flush_ready = (flush_ready_fn)flush_ready_p;
}

void lvgpu_lvgl_set_color_16_swap(int swap) { g_swap = swap ? 1 : 0; }

void lvgpu_lvgl_set_deferred(int on) { g_deferred = on > 0 ? 1 : (on < 0 ? -1 : 0); }

void lvgpu_lvgl_inject_failure(int nth_blend) { g_inject_fail = nth_blend > 0 ? nth_blend : 0; }

void * lvgpu_lvgl_buf_alloc(size_t bytes)
{
    void * p = nullptr;
    if(bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return p;
}
void lvgpu_lvgl_buf_free(void * p)
{
    if(p) (void)hipHostFree(p);
}

size_t lvgpu_draw_ctx_size(void) { return sizeof(ShimCtx); }

void lvgpu_draw_ctx_init(lvgpu_lv_disp_drv_t * drv, lvgpu_lv_draw_ctx_t * draw_ctx)
{
    if(g_sw_init_ctx == nullptr) g_sw_init_ctx = (init_ctx_fn)lookup("lv_draw_sw_init_ctx");
    if(g_sw_blend_basic == nullptr) g_sw_blend_basic = (blend_fn)lookup("lv_draw_sw_blend_basic");
    ShimCtx * s = reinterpret_cast<ShimCtx *>(draw_ctx);
    if(g_sw_init_ctx == nullptr) {
        fprintf(stderr, "lvgpu: lv_draw_sw_init_ctx not bound (call lvgpu_lvgl_bind) -- aborting\n");
        abort();
    }
    g_sw_init_ctx(drv, draw_ctx);  // fills the primitive table and zeroes lv_draw_sw_ctx_t (lv_draw_sw.c:40-65)
    memset(reinterpret_cast<char *>(s) + sizeof(lvgpu_lv_draw_ctx_t) + sizeof(blend_fn), 0,
           sizeof(ShimCtx) - sizeof(lvgpu_lv_draw_ctx_t) - sizeof(blend_fn));
    s->magic = MAGIC;
    s->drv = drv;
    s->blend = lvgpu_blend;
    s->base_draw.wait_for_finish = lvgpu_wait_for_finish;
    int dev = -1;
    HIP_OR_DIE(hipGetDevice(&dev));  // the calling thread's current device; fails when there is none
    HIP_OR_DIE(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    // LVGL's own software rotation reads the draw buffer on the CPU between rendering and flush_cb
    // (libraries/lvgl/src/core/lv_refr.c:1262-1296) with no hook in between once wait_for_finish is cleared, so a
    // display that uses it keeps the immediate mode.
    const bool deferred = g_deferred > 0 || (g_deferred < 0 && drv->flush_cb == lvgpu_flush_cb);
    if(deferred && !drv->sw_rotate) {
        Queue * q = new Queue();
        s->q = q;
        // lv_draw_sw_blend calls wait_for_finish before EVERY blend (lv_draw_sw_blend.c:103), which would submit
        // the queue item by item; every caller checks the pointer, so it is cleared and the hooks below submit.
        s->base_draw.wait_for_finish = nullptr;
        q->sw_layer_init = (layer_init_fn)s->base_draw.layer_init;
        q->sw_layer_adjust = (layer_adjust_fn)s->base_draw.layer_adjust;
        q->sw_layer_blend = (layer_blend_fn)s->base_draw.layer_blend;
        q->sw_layer_destroy = (layer_destroy_fn)s->base_draw.layer_destroy;
        q->sw_buffer_copy = (buffer_copy_fn)s->base_draw.buffer_copy;
        s->base_draw.layer_init = (void *)hook_layer_init;
        s->base_draw.layer_adjust = (void *)hook_layer_adjust;
        s->base_draw.layer_blend = (void *)hook_layer_blend;
        s->base_draw.layer_destroy = (void *)hook_layer_destroy;
        s->base_draw.buffer_copy = hook_buffer_copy;
    }
}

void lvgpu_draw_ctx_deinit(lvgpu_lv_disp_drv_t * drv, lvgpu_lv_draw_ctx_t * draw_ctx)
{
    (void)drv;
    ShimCtx * s = reinterpret_cast<ShimCtx *>(draw_ctx);
    if(s->magic == MAGIC) {
        if(s->q) {
            submit(s);  // e.g. a canvas or snapshot drawn through a temporary context expects its pixels now
            if(s->stream) (void)hipStreamSynchronize(s->stream);
            if(s->q->compiled) (void)lvgpu_list_destroy(s->q->compiled);
            s->q->arena.release();
            if(s->q->d_fb) (void)hipFree(s->q->d_fb);
            if(s->q->d_arena) (void)hipFree(s->q->d_arena);
            delete s->q;
        }
        if(s->stream) (void)hipStreamSynchronize(s->stream);
        if(s->d_dst) (void)hipFree(s->d_dst);
        if(s->d_src) (void)hipFree(s->d_src);
        if(s->d_mask) (void)hipFree(s->d_mask);
        if(s->stream) (void)hipStreamDestroy(s->stream);
    }
    memset(s, 0, sizeof(ShimCtx));  // lv_draw_sw_deinit_ctx zeroes the context too (lv_draw_sw.c:67-73)
}

void lvgpu_wait_for_finish(lvgpu_lv_draw_ctx_t * draw_ctx)
{
    ShimCtx * s = reinterpret_cast<ShimCtx *>(draw_ctx);
    if(s == nullptr || s->magic != MAGIC) return;
    if(s->q) submit(s);
    if(s->stream) HIP_OR_DIE(hipStreamSynchronize(s->stream));
}

void lvgpu_blend(lvgpu_lv_draw_ctx_t * draw_ctx, const lvgpu_lv_draw_sw_blend_dsc_t * dsc)
{
    ShimCtx * s = reinterpret_cast<ShimCtx *>(draw_ctx);
    lvgpu_lv_disp_drv_t * drv = s->drv;

    // A request outside this library's coverage goes to the host's own lv_draw_sw_blend_basic, the convention of
    // every LVGL GPU back-end (libraries/lvgl/src/draw/stm32_dma2d/lv_gpu_stm32_dma2d.c:127-130,207): a display
    // with a custom set_px_cb, whose pixel format only that callback knows.
    if(drv->set_px_cb != nullptr) {
        if(s->q) submit(s);
        g_blend_delegated++;
        if(g_sw_blend_basic) g_sw_blend_basic(draw_ctx, dsc);
        else {
            fprintf(stderr, "lvgpu: request needs lv_draw_sw_blend_basic, which is not bound -- aborting\n");
            abort();
        }
        return;
    }

    // lv_draw_sw_blend_basic, libraries/lvgl/src/draw/sw/lv_draw_sw_blend.c:111-166
    uint8_t * mask;
    if(dsc->mask_buf && dsc->mask_res == LVGPU_MASK_RES_TRANSP) return;
    else if(dsc->mask_res == LVGPU_MASK_RES_FULL_COVER) mask = nullptr;
    else mask = dsc->mask_buf;

    const int dest_stride = aw(draw_ctx->buf_area);
    lvgpu_lv_area_t a;
    a.x1 = dsc->blend_area->x1 > draw_ctx->clip_area->x1 ? dsc->blend_area->x1 : draw_ctx->clip_area->x1;
    a.y1 = dsc->blend_area->y1 > draw_ctx->clip_area->y1 ? dsc->blend_area->y1 : draw_ctx->clip_area->y1;
    a.x2 = dsc->blend_area->x2 < draw_ctx->clip_area->x2 ? dsc->blend_area->x2 : draw_ctx->clip_area->x2;
    a.y2 = dsc->blend_area->y2 < draw_ctx->clip_area->y2 ? dsc->blend_area->y2 : draw_ctx->clip_area->y2;
    if(a.x1 > a.x2 || a.y1 > a.y2) return;
    const int w = aw(&a), h = ah(&a);

    const size_t dbpp = drv->screen_transp ? 3 : 2;  // lv_draw_sw_blend.c:126-133
    uint8_t * dest = reinterpret_cast<uint8_t *>(draw_ctx->buf) +
                     ((size_t)dest_stride * (a.y1 - draw_ctx->buf_area->y1) + (a.x1 - draw_ctx->buf_area->x1)) * dbpp;
    const uint16_t * src = dsc->src_buf;
    int src_stride = 0;
    if(src) {
        src_stride = aw(dsc->blend_area);
        src += (size_t)src_stride * (a.y1 - dsc->blend_area->y1) + (a.x1 - dsc->blend_area->x1);
    }
    int mask_stride = 0;
    // :150-156: with anti-aliasing off the mask is rounded to 0 / 255 -- in the caller's buffer, over the whole
    // mask_area, as the reference does it: LVGL's producers read their masks again after the call (the shadow renderer
    // blends straight out of its shadow buffer, lv_draw_sw_rect.c:572, and later takes opacities from the same bytes,
    // :661), so the rounding is part of what a drop-in has to reproduce.  The batched ABI has the flag
    // LVGPU_BLEND_FLAG_NO_AA for callers whose masks are theirs alone; here the bytes are rounded already.
    if(mask && drv->antialiasing == 0) {
        const int32_t n = aw(dsc->mask_area) * ah(dsc->mask_area);
        for(int32_t i = 0; i < n; i++) mask[i] = mask[i] > 128 ? 255 : 0;
    }
    const uint8_t no_aa = 0;
    if(mask) {
        mask_stride = aw(dsc->mask_area);
        mask += (size_t)mask_stride * (a.y1 - dsc->mask_area->y1) + (a.x1 - dsc->mask_area->x1);
    }

    const size_t px = (size_t)w * h;
    if(s->q) {  // deferred: keep private copies of the rectangles, draw later
        Queue * q = s->q;
        const lvgpu_area_t ba{draw_ctx->buf_area->x1, draw_ctx->buf_area->y1, draw_ctx->buf_area->x2, draw_ctx->buf_area->y2};
        const bool transp = drv->screen_transp != 0;
        if(!q->items.empty() && (q->buf != draw_ctx->buf || memcmp(&q->buf_area, &ba, sizeof(ba)) != 0 || q->transp != transp))
            submit(s);  // the context was pointed at another buffer (layers do that)
        if(q->items.empty()) q->covered = false;
        q->buf = draw_ctx->buf;
        q->buf_area = ba;
        q->transp = transp;
        QItem it;
        memset(&it, 0, sizeof(it));  // padding too: items are compared bytewise between refreshes
        it.area = lvgpu_area_t{a.x1, a.y1, a.x2, a.y2};
        it.has_src = src != nullptr;
        it.has_mask = mask != nullptr;
        it.color = dsc->color;
        it.opa = dsc->opa;
        it.mode = dsc->blend_mode;
        it.flags = no_aa;
        if(src) {
            it.src_off = (q->arena.size + 15) & ~(size_t)15;
            uint8_t * base = q->arena.grow_to(it.src_off + px * 2);
            for(int y = 0; y < h; y++) memcpy(base + it.src_off + (size_t)y * w * 2, src + (size_t)y * src_stride, (size_t)w * 2);
        }
        if(mask) {
            it.mask_off = (q->arena.size + 15) & ~(size_t)15;
            uint8_t * base = q->arena.grow_to(it.mask_off + px);
            if(transp && src == nullptr && dsc->opa >= 253)
                memcpy(base + it.mask_off, mask, px);  // fill_argb's opaque masked loop: one run of w*h bytes
            else
                for(int y = 0; y < h; y++) memcpy(base + it.mask_off + (size_t)y * w, mask + (size_t)y * mask_stride, (size_t)w);
        }
        // An item that replaces every pixel of the buffer whatever it held (fill_normal :254-259 / map_normal :655-660
        // over the whole buffer; with a transparent screen fill_argb :465-477 / map_argb :768-790): what the buffer
        // holds now never reaches the result, so it need not travel to the device.
        if(mask == nullptr && dsc->opa >= 253 && (dsc->blend_mode == 0 || (transp && src == nullptr)) && a.x1 == ba.x1 && a.y1 == ba.y1 &&
           a.x2 == ba.x2 && a.y2 == ba.y2)
            q->covered = true;
        q->items.push_back(it);
        g_blend_gpu++;
        return;
    }
    // Immediate mode.  Until the result starts to travel back the host buffer is untouched, so a failure up to that
    // point (no memory for the staging buffers, a refused copy or launch) is counted and the call is handed to the host
    // program's lv_draw_sw_blend_basic -- the convention of LVGL's GPU back-ends
    // (libraries/lvgl/src/draw/stm32_dma2d/lv_gpu_stm32_dma2d.c:127-130).  Without a bound fallback, or once the copy
    // back has been issued, a failure ends the program: a frame with a hole in it must not go unnoticed.
    hipStream_t st = s->stream;
    const char * what = nullptr;
    hipError_t e = hipSuccess;
    int rc = LVGPU_OK;
#define STEP(expr)                                                       \
    do {                                                                 \
        if(what == nullptr && (e = (expr)) != hipSuccess) what = #expr; \
    } while(0)
    if(g_inject_fail > 0 && --g_inject_fail == 0) {  // test hook: see lvgpu_lvgl_inject_failure
        what = "injected failure";
        e = hipErrorOutOfMemory;
    }
    if(what == nullptr && px > s->cap_px) {
        if(s->d_dst) (void)hipFree(s->d_dst);
        if(s->d_src) (void)hipFree(s->d_src);
        if(s->d_mask) (void)hipFree(s->d_mask);
        s->d_dst = s->d_src = s->d_mask = nullptr;
        s->cap_px = 0;
        const size_t cap = px < 65536 ? 65536 : px;
        STEP(hipMalloc(&s->d_dst, cap * 3));  // 3 bytes per pixel when the screen is transparent (ARGB8565)
        STEP(hipMalloc(&s->d_src, cap * 2));
        STEP(hipMalloc(&s->d_mask, cap));
        if(what == nullptr) s->cap_px = cap;
    }
    const bool overwrite = mask == nullptr && dsc->opa >= 253 && dsc->blend_mode == 0 && !drv->screen_transp;
    if(!overwrite)  // every other case reads the destination
        STEP(hipMemcpy2DAsync(s->d_dst, (size_t)w * dbpp, dest, (size_t)dest_stride * dbpp, (size_t)w * dbpp, h, hipMemcpyHostToDevice, st));
    if(src) STEP(hipMemcpy2DAsync(s->d_src, (size_t)w * 2, src, (size_t)src_stride * 2, (size_t)w * 2, h, hipMemcpyHostToDevice, st));
    if(mask) {
        if(drv->screen_transp && src == nullptr && dsc->opa >= 253)
            // fill_argb's opaque masked loop walks the mask as ONE run of w*h bytes (lv_draw_sw_blend.c:494-505: the
            // pointer is never advanced by mask_stride - w), so that run is what has to be staged
            STEP(hipMemcpyAsync(s->d_mask, mask, px, hipMemcpyHostToDevice, st));
        else
            STEP(hipMemcpy2DAsync(s->d_mask, (size_t)w, mask, (size_t)mask_stride, (size_t)w, h, hipMemcpyHostToDevice, st));
    }
    if(what == nullptr) {
        lvgpu_blend_batch_t b;
        memset(&b, 0, sizeof(b));
        b.dst = s->d_dst;
        b.dst_frame_stride = (int64_t)px;
        b.buf_area = lvgpu_area_t{0, 0, (int16_t)(w - 1), (int16_t)(h - 1)};
        b.clip_area = b.buf_area;
        b.blend_area = b.buf_area;
        b.mask_area = b.buf_area;
        b.src = src ? s->d_src : nullptr;
        b.src_format = src ? LVGPU_SRC_RGB565 : LVGPU_SRC_FILL;
        b.mask = mask ? reinterpret_cast<const uint8_t *>(s->d_mask) : nullptr;
        b.mask_res = LVGPU_MASK_RES_CHANGED;
        b.n_frames = 1;
        b.color = dsc->color;
        b.opa = dsc->opa;
        b.blend_mode = dsc->blend_mode;
        b.color_16_swap = (uint8_t)g_swap;
        b.dst_argb8565 = drv->screen_transp ? 1 : 0;
        b.flags = no_aa;
        rc = lvgpu_blend_batch(&b, st);
        if(rc != LVGPU_OK) what = "lvgpu_blend_batch";
    }
#undef STEP
    if(what != nullptr) {
        g_errors++;
        (void)hipGetLastError();
        (void)hipStreamSynchronize(st);  // uploads that did start read src / mask, which the caller reuses after this call
        if(g_sw_blend_basic == nullptr) {
            fprintf(stderr, "lvgpu: %s failed (%s) and lv_draw_sw_blend_basic is not bound -- aborting\n", what,
                    rc != LVGPU_OK ? lvgpu_last_error() : hipGetErrorString(e));
            abort();
        }
        static std::atomic<int> said{0};
        if(said.fetch_add(1) < 5)
            fprintf(stderr, "lvgpu: %s failed (%s); this blend is drawn by lv_draw_sw_blend_basic\n", what,
                    rc != LVGPU_OK ? lvgpu_last_error() : hipGetErrorString(e));
        g_blend_delegated++;
        g_sw_blend_basic(draw_ctx, dsc);
        return;
    }
    HIP_OR_DIE(hipMemcpy2DAsync(dest, (size_t)dest_stride * dbpp, s->d_dst, (size_t)w * dbpp, (size_t)w * dbpp, h,
                                hipMemcpyDeviceToHost, st));
    // src_buf / mask_buf may be reused by the caller as soon as we return, and the CPU reads the draw buffer.
    HIP_OR_DIE(hipStreamSynchronize(st));
    g_blend_gpu++;
}

void lvgpu_flush_cb(lvgpu_lv_disp_drv_t * drv, const lvgpu_lv_area_t * area, lvgpu_lv_color_t * color_p)
{
    const lvgpu_flush_cfg_t * cfg = reinterpret_cast<const lvgpu_flush_cfg_t *>(drv->user_data);
    if(cfg == nullptr || cfg->draw_bitmap == nullptr) {
        fprintf(stderr, "lvgpu_flush_cb: drv->user_data must point to a lvgpu_flush_cfg_t with a draw_bitmap sink\n");
        abort();
    }
    // only a context this library initialised carries a queue (deferred mode)
    ShimCtx * s = (drv->draw_ctx && drv->draw_ctx_init == lvgpu_draw_ctx_init) ? shim_of(drv->draw_ctx) : nullptr;
    const int x_start = area->x1, x_end = area->x2, y_start = area->y1, y_end = area->y2;
    const int W = x_end - x_start + 1, H = y_end - y_start + 1;
    const int rot = cfg->sw_rotate;

    if(cfg->trans_size == 0) {  // src/lv_port.c:568-570
        if(s) lvgpu_wait_for_finish(drv->draw_ctx);  // the queued blends are drawn now
        cfg->draw_bitmap(cfg->user, x_start, y_start, x_end + 1, y_end + 1, color_p);
        flush_ready(drv);
        return;
    }
    const bool transposing = rot == LVGPU_ROT_90 || rot == LVGPU_ROT_270;
    const int step = transposing ? (int)(cfg->trans_size / (uint32_t)H) : (int)(cfg->trans_size / (uint32_t)W);
    if(step <= 0) {
        fprintf(stderr, "lvgpu_flush_cb: trans_size %u is smaller than one line of the flushed area\n", cfg->trans_size);
        abort();
    }

    // rotate the whole area on the device (staging buffers are kept between calls and grown on demand)
    const size_t px = (size_t)W * H;
    struct FlushStaging {
        void *d_in = nullptr, *d_out = nullptr;
        uint16_t * panel = nullptr;
        size_t cap = 0;
        ~FlushStaging()  // at thread exit (every use is synchronous, nothing can still be in flight)
        {
            if(d_in) (void)hipFree(d_in);
            if(d_out) (void)hipFree(d_out);
            if(panel) (void)hipHostFree(panel);
        }
    };
    static thread_local FlushStaging stg;
    if(px > stg.cap) {
        if(stg.d_in) HIP_OR_DIE(hipFree(stg.d_in));
        if(stg.d_out) HIP_OR_DIE(hipFree(stg.d_out));
        if(stg.panel) HIP_OR_DIE(hipHostFree(stg.panel));
        HIP_OR_DIE(hipMalloc(&stg.d_in, px * 2));
        HIP_OR_DIE(hipMalloc(&stg.d_out, px * 2));
        HIP_OR_DIE(hipHostMalloc(reinterpret_cast<void **>(&stg.panel), px * 2, hipHostMallocDefault));
        stg.cap = px;
    }
    void *d_in = stg.d_in, *d_out = stg.d_out;
    uint16_t * panel = stg.panel;
    // Deferred mode, the whole draw buffer is flushed (full_refresh, the reference's configuration, src/lv_port.c:242)
    // and it is the buffer the queue draws into: the launch that draws the queue writes the panel image as well, and
    // the buffer travels to the device once instead of twice.
    Queue * q = s ? s->q : nullptr;
    const bool fused = q && !q->items.empty() && !q->transp && q->buf == (void *)color_p && q->buf_area.x1 == x_start &&
                       q->buf_area.y1 == y_start && q->buf_area.x2 == x_end && q->buf_area.y2 == y_end;
    if(fused) {
        const FlushReq fr{d_out, panel, rot, cfg->swap_bytes};
        submit(s, &fr);
    }
    else {
        if(s) lvgpu_wait_for_finish(drv->draw_ctx);  // the queued blends are drawn now
        HIP_OR_DIE(hipMemcpy(d_in, color_p, px * 2, hipMemcpyHostToDevice));
        lvgpu_flush_batch_t f;
        memset(&f, 0, sizeof(f));
        f.in = d_in;
        f.out = d_out;
        f.in_frame_stride = f.out_frame_stride = (int64_t)px;
        f.width = W;
        f.height = H;
        f.rotation = rot;
        f.swap_bytes = cfg->swap_bytes;
        f.n_frames = 1;
        int rc = lvgpu_flush_rotate_batch(&f, nullptr);
        if(rc != LVGPU_OK) {
            fprintf(stderr, "lvgpu: lvgpu_flush_rotate_batch failed (%d): %s -- aborting\n", rc, lvgpu_last_error());
            abort();
        }
        HIP_OR_DIE(hipMemcpy(panel, d_out, px * 2, hipMemcpyDeviceToHost));
    }

    // Hand the panel image to the sink in the reference's transport chunks (src/lv_port.c:466-566): same
    // rectangles, same order.  Each chunk is a contiguous slice of the rotated image.
    const lvgpu_area_t fa{(int16_t)x_start, (int16_t)y_start, (int16_t)x_end, (int16_t)y_end};
    const int n_chunks = lvgpu_flush_chunks(&fa, drv->hor_res, drv->ver_res, rot, cfg->trans_size, nullptr, 0);
    static thread_local std::vector<lvgpu_flush_chunk_t> chunks;
    chunks.resize(n_chunks > 0 ? (size_t)n_chunks : 0);
    if(n_chunks <= 0 || lvgpu_flush_chunks(&fa, drv->hor_res, drv->ver_res, rot, cfg->trans_size, chunks.data(), n_chunks) != n_chunks) {
        fprintf(stderr, "lvgpu_flush_cb: %s -- aborting\n", lvgpu_last_error());
        abort();
    }
    for(int i = 0; i < n_chunks; i++) {
        if(i == 0 && cfg->draw_wait) cfg->draw_wait(cfg->user);  // :548-551
        const lvgpu_flush_chunk_t & c = chunks[(size_t)i];
        cfg->draw_bitmap(cfg->user, c.x_start, c.y_start, c.x_end, c.y_end, panel + c.offset_px);
    }
    g_flush_gpu++;
    flush_ready(drv);  // :571
}

int lvgpu_rotation_for_lvgl_sw_rotate(int lv_disp_rot)
{
    switch(lv_disp_rot) {
        case 0: return LVGPU_ROT_NONE;
        case 1: return LVGPU_ROT_270;
        case 2: return LVGPU_ROT_180;
        case 3: return LVGPU_ROT_90;
        default: return -1;
    }
}

void lvgpu_lvgl_get_stats(lvgpu_lvgl_stats_t * out)
{
    out->blend_gpu = g_blend_gpu;
    out->blend_delegated = g_blend_delegated;
    out->flush_gpu = g_flush_gpu;
    out->errors = g_errors;
    out->blend_lists = g_lists;
    out->list_replays = g_replays;
    out->submit_ns = g_submit_ns;
    out->lists_in_place = g_in_place;
}
void lvgpu_lvgl_reset_stats(void)
{
    g_blend_gpu = 0;
    g_blend_delegated = 0;
    g_flush_gpu = 0;
    g_errors = 0;
    g_lists = 0;
    g_replays = 0;
    g_submit_ns = 0;
    g_in_place = 0;
}

}  // extern "C"
