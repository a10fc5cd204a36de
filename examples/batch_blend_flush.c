/*
 * batch_blend_flush.c -- a plain C host program on the C ABI (include/lvgpu.h only; no HIP, LVGL or Python).
 *
 * Builds N synthetic 480x320 RGB565 draw buffers and ARGB8888 images in device memory, draws every image over its
 * buffer (lv_draw_sw_blend semantics) fused with the 90-degree flush rotation, and prints order-independent
 * checksums of the blended draw buffers and of the panel images; then it draws a small display list (background,
 * 40 one-row masked fills, a translucent panel) with lvgpu_blend_list_flush_batch in one launch.
 * tests/test_gpu_c_host.py runs it and compares the checksums with the CPU oracle's.
 *
 *   gcc -O2 -Iinclude examples/batch_blend_flush.c -Lvvc_amd/lib -llvgpu -Wl,-rpath,'$ORIGIN/../vvc_amd/lib' -o examples/batch_blend_flush
 */
#include <inttypes.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "lvgpu.h"

#define CHECK(call)                                                                      \
    do {                                                                                 \
        int rc_ = (call);                                                                \
        if(rc_ != LVGPU_OK) {                                                            \
            fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, lvgpu_last_error());     \
            return 1;                                                                    \
        }                                                                                \
    } while(0)

int main(int argc, char ** argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 64;
    const int rotation = argc > 2 ? atoi(argv[2]) : LVGPU_ROT_90;
    const int W = 480, H = 320;
    const size_t px = (size_t)W * H, total = px * (size_t)n;

    CHECK(lvgpu_init(0));
    void *dst, *src, *panel;
    uint64_t * sums;
    CHECK(lvgpu_malloc(&dst, total * 2));
    CHECK(lvgpu_malloc(&src, total * 4));
    CHECK(lvgpu_malloc(&panel, total * 2));
    CHECK(lvgpu_malloc((void **)&sums, 2 * sizeof(uint64_t)));
    CHECK(lvgpu_memset(sums, 0, 2 * sizeof(uint64_t), NULL));
    CHECK(lvgpu_synth(dst, LVGPU_SYNTH_U16, 0x5EED0001u, 0, total, NULL));
    CHECK(lvgpu_synth(src, LVGPU_SYNTH_ARGB_UI, 0x5EED0002u, 0, total, NULL));

    lvgpu_blend_flush_batch_t d;
    memset(&d, 0, sizeof(d));
    d.blend.dst = dst;
    d.blend.dst_frame_stride = (int64_t)px;
    d.blend.buf_area = (lvgpu_area_t){0, 0, (int16_t)(W - 1), (int16_t)(H - 1)};
    d.blend.clip_area = d.blend.buf_area;
    d.blend.blend_area = d.blend.buf_area;
    d.blend.src = src;
    d.blend.src_frame_stride = (int64_t)px;
    d.blend.src_format = LVGPU_SRC_ARGB8888;
    d.blend.opa = 255;
    d.blend.color_16_swap = 1;
    d.blend.n_frames = n;
    d.out = panel;
    d.out_frame_stride = (int64_t)px;
    d.rotation = rotation;
    CHECK(lvgpu_blend_flush_batch(&d, NULL));

    CHECK(lvgpu_checksum(dst, 2, 0, total, sums + 0, NULL));
    CHECK(lvgpu_checksum(panel, 2, 0, total, sums + 1, NULL));
    uint64_t h[2];
    CHECK(lvgpu_memcpy_d2h(h, sums, sizeof(h), NULL));
    CHECK(lvgpu_stream_synchronize(NULL));
    printf("frames %d rotation %d kernel %s\n", n, rotation, lvgpu_last_kernel());
    printf("draw_buffers %" PRIu64 "\npanel_images %" PRIu64 "\n", h[0], h[1]);

    /* A display list on top: an opaque background, 40 one-row masked fills (what LVGL's producers emit for a rounded
     * shape) and a translucent panel, applied in order in ONE launch together with the flush. */
    enum { N_ROWS = 40, N_ITEMS = N_ROWS + 2 };
    void * mask;
    CHECK(lvgpu_malloc(&mask, 200 * N_ROWS));
    CHECK(lvgpu_synth(mask, LVGPU_SYNTH_MASK_UI, 0x5EED0003u, 0, 200 * N_ROWS, NULL));
    static lvgpu_blend_batch_t items[N_ITEMS];
    for(int i = 0; i < N_ITEMS; i++) {
        items[i] = d.blend;
        items[i].src = NULL;
        items[i].src_format = LVGPU_SRC_FILL;
    }
    items[0].color = 0x1084; /* background, opa 255 */
    for(int r = 0; r < N_ROWS; r++) {
        lvgpu_blend_batch_t * it = &items[1 + r];
        it->blend_area = (lvgpu_area_t){37, (int16_t)(50 + r), 236, (int16_t)(50 + r)};
        it->mask_area = it->blend_area;
        it->mask = (const uint8_t *)mask + 200 * r;
        it->mask_res = LVGPU_MASK_RES_CHANGED;
        it->color = 0xE0FF;
        it->opa = 230;
    }
    items[N_ITEMS - 1].blend_area = (lvgpu_area_t){100, 20, 400, 300};
    items[N_ITEMS - 1].color = 0x00F8;
    items[N_ITEMS - 1].opa = 96;
    lvgpu_list_plan_t plan;
    CHECK(lvgpu_blend_list_plan(NULL, items, N_ITEMS, 1, rotation, &plan));
    CHECK(lvgpu_blend_list_flush_batch(&items[0], items, N_ITEMS, panel, (int64_t)px, rotation, 0, NULL));
    CHECK(lvgpu_memset(sums, 0, 2 * sizeof(uint64_t), NULL));
    CHECK(lvgpu_checksum(dst, 2, 0, total, sums + 0, NULL));
    CHECK(lvgpu_checksum(panel, 2, 0, total, sums + 1, NULL));
    CHECK(lvgpu_memcpy_d2h(h, sums, sizeof(h), NULL));
    CHECK(lvgpu_stream_synchronize(NULL));
    printf("list_kernel %s\nlist_launches %d\nlist_steps %lld\n", lvgpu_last_kernel(), plan.launches,
           (long long)(plan.item_steps + plan.batch_steps));
    printf("list_draw_buffers %" PRIu64 "\nlist_panel_images %" PRIu64 "\n", h[0], h[1]);
    CHECK(lvgpu_free(mask));

    /* error behaviour: an area to draw that reaches outside the buffer is rejected, nothing is drawn */
    d.blend.clip_area.x2 = (int16_t)(W + 10);
    d.blend.blend_area.x2 = (int16_t)(W + 10);
    int rc = lvgpu_blend_flush_batch(&d, NULL);
    printf("bad_clip_rc %d\n", rc);

    CHECK(lvgpu_free(dst));
    CHECK(lvgpu_free(src));
    CHECK(lvgpu_free(panel));
    CHECK(lvgpu_free(sums));
    return rc == LVGPU_ERR_ARG ? 0 : 2;
}
