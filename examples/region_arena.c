/*
 * region_arena.c -- a plain C host that places its buffers by region of the MI355X's device memory.
 *
 * The 288 GB behave as three regions of 96 GB; the fused blend + quarter turn runs up to 9 % faster when the draw
 * buffers, the sources and the panel images lie in different regions.  A process cannot ask for physical addresses,
 * so this host takes nearly all device memory in one allocation, finds the regions inside it by timing (the launch
 * with its panel images moved through the arena in 8-GB steps: slow where they share the draw buffers' region), and
 * then runs the batch in both placements: results identical, rates printed.
 *
 *   examples/region_arena [n_frames = 4096]
 *
 * Prints   map <one digit per 8-GB slot>, by_region_ms / one_region_ms (per launch) and their % of 8 TB/s, and the
 * checksums of draw buffers and panel images after one launch on fresh inputs in either placement.
 * tests/test_gpu_c_host.py runs it and compares the checksums with the CPU oracle's.
 */
#define _POSIX_C_SOURCE 199309L
#include <inttypes.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "lvgpu.h"

#define CHECK(call)                                                                      \
    do {                                                                                 \
        int rc_ = (call);                                                                \
        if(rc_ != LVGPU_OK) {                                                            \
            fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, lvgpu_last_error());     \
            return 1;                                                                    \
        }                                                                                \
    } while(0)

enum { W = 480, H = 320, PROBE_FRAMES = 2048, MAX_SLOTS = 64 };
static const size_t GB = (size_t)1 << 30, SLOT = (size_t)8 << 30;
static const size_t PX = (size_t)W * H;

static double now_ms(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec * 1e3 + t.tv_nsec * 1e-6;
}

static lvgpu_blend_flush_batch_t batch(void * dst, void * src, void * panel, int n)
{
    lvgpu_blend_flush_batch_t d;
    memset(&d, 0, sizeof(d));
    d.blend.dst = dst;
    d.blend.dst_frame_stride = (int64_t)PX;
    d.blend.buf_area = (lvgpu_area_t){0, 0, W - 1, H - 1};
    d.blend.clip_area = d.blend.blend_area = d.blend.buf_area;
    d.blend.src = src;
    d.blend.src_frame_stride = (int64_t)PX;
    d.blend.src_format = LVGPU_SRC_ARGB8888;
    d.blend.opa = 255;
    d.blend.color_16_swap = 1;
    d.blend.n_frames = n;
    d.out = panel;
    d.out_frame_stride = (int64_t)PX;
    d.rotation = LVGPU_ROT_90;
    return d;
}

/* milliseconds per launch, `reps` launches after one untimed; negative on failure */
static double time_launch(const lvgpu_blend_flush_batch_t * d, int reps)
{
    if(lvgpu_blend_flush_batch(d, NULL) != LVGPU_OK || lvgpu_stream_synchronize(NULL) != LVGPU_OK) return -1.0;
    const double t0 = now_ms();
    for(int i = 0; i < reps; i++)
        if(lvgpu_blend_flush_batch(d, NULL) != LVGPU_OK) return -1.0;
    if(lvgpu_stream_synchronize(NULL) != LVGPU_OK) return -1.0;
    return (now_ms() - t0) / reps;
}

/* One sweep: draw buffers and sources in slot `anchor`, panel images in the upper half of every slot with want[s] set.
 * slow[s] = 1 where the launch took longer than the geometric mean of the extremes; returns 0 when the extremes are
 * within 4 % of each other (no structure), -1 on failure, 1 otherwise. */
static int sweep(char * arena, int n_slots, int anchor, const char * want, char * slow)
{
    double t[MAX_SLOTS], lo = 1e30, hi = 0.0;
    for(int s = 0; s < n_slots; s++) {
        if(!want[s]) continue;
        lvgpu_blend_flush_batch_t d = batch(arena + anchor * SLOT, arena + anchor * SLOT + 1 * GB,
                                            arena + s * SLOT + 4 * GB, PROBE_FRAMES);
        t[s] = time_launch(&d, 3);
        if(t[s] < 0.0) return -1;
        if(t[s] < lo) lo = t[s];
        if(t[s] > hi) hi = t[s];
    }
    if(hi < 1.04 * lo) return 0;
    for(int s = 0; s < n_slots; s++)
        if(want[s]) slow[s] = t[s] * t[s] >= lo * hi;
    return 1;
}

/* start of the longest run of slots labelled r (its second slot where the run has three or more), or -1 */
static int region_start(const char * label, int n_slots, int r, int * len_out)
{
    int best = -1, best_len = 0;
    for(int i = 0; i < n_slots;) {
        int j = i;
        while(j + 1 < n_slots && label[j + 1] == label[i]) j++;
        if(label[i] == r && j - i + 1 > best_len) best = i, best_len = j - i + 1;
        i = j + 1;
    }
    if(len_out) *len_out = best_len >= 3 ? best_len - 2 : best_len;
    return best_len >= 3 ? best + 1 : best;
}

int main(int argc, char ** argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 4096;
    const size_t total = PX * (size_t)n;
    if(n < 1 || total * 4 > 5 * SLOT) {
        fprintf(stderr, "n_frames 1 .. %zu\n", 5 * SLOT / (PX * 4));
        return 2;
    }
    CHECK(lvgpu_init(0));

    /* the arena: the largest multiple of 8 GB the device grants, less 16 GB for everybody else */
    char * arena = NULL;
    size_t arena_bytes = 0;
    for(size_t gb = 288; gb >= 64 && !arena; gb -= 8) {
        void * p;
        if(lvgpu_malloc(&p, gb * GB) != LVGPU_OK) continue;
        CHECK(lvgpu_free(p));
        if(gb < 64 + 16) break;
        arena_bytes = (gb - 16) * GB;
        CHECK(lvgpu_malloc(&p, arena_bytes));
        arena = p;
    }
    if(!arena) {
        fprintf(stderr, "less than 80 GB of device memory free: no arena\n");
        return 3;
    }
    const int n_slots = (int)(arena_bytes / SLOT) < MAX_SLOTS ? (int)(arena_bytes / SLOT) : MAX_SLOTS;

    /* the region map: sweep 1 with the anchor in slot 0 finds slot 0's region; sweep 2 with the anchor in the first
     * slot outside it splits the rest */
    char label[MAX_SLOTS] = {0}, want[MAX_SLOTS], slow[MAX_SLOTS] = {0};
    memset(want, 1, sizeof(want));
    int rc = sweep(arena, n_slots, 0, want, slow);
    if(rc < 0) {
        fprintf(stderr, "probe launch failed: %s\n", lvgpu_last_error());
        return 1;
    }
    if(rc == 1) {
        int first_other = -1;
        for(int s = 0; s < n_slots; s++) {
            want[s] = !slow[s];
            if(want[s] && first_other < 0) first_other = s;
        }
        char slow2[MAX_SLOTS] = {0};
        rc = sweep(arena, n_slots, first_other, want, slow2);
        if(rc < 0) return 1;
        for(int s = 0; s < n_slots; s++)
            if(want[s]) label[s] = (rc == 1 && !slow2[s]) ? 2 : 1;
    }
    printf("arena_gb %zu\nmap ", arena_bytes / GB);
    for(int s = 0; s < n_slots; s++) putchar('0' + label[s]);
    putchar('\n');

    /* placements: one buffer per region (a region the map does not have: the next slots of region 0), and all three
     * one after the other at the start of region 0 */
    int len0, len1, len2;
    const int s0 = region_start(label, n_slots, 0, &len0);
    int s1 = region_start(label, n_slots, 1, &len1), s2 = region_start(label, n_slots, 2, &len2);
    if(s1 < 0) s1 = s0 + 2;
    if(s2 < 0) s2 = s1 + 3;
    if(total * 8 > (size_t)len0 * SLOT || total * 4 > (size_t)(len1 > 0 ? len1 : 1) * SLOT || total * 2 > (size_t)(len2 > 0 ? len2 : 1) * SLOT || s2 + 1 > n_slots) {
        fprintf(stderr, "the regions found are too small for %d frames\n", n);
        return 4;
    }
    char *dst[2] = {arena + s0 * SLOT, arena + s0 * SLOT}, *src[2] = {arena + s1 * SLOT, arena + s0 * SLOT + total * 2},
         *pan[2] = {arena + s2 * SLOT, arena + s0 * SLOT + total * 6};
    static const char * const name[2] = {"by_region", "one_region"};
    uint64_t * sums;
    CHECK(lvgpu_malloc((void **)&sums, 2 * sizeof(uint64_t)));
    for(int k = 0; k < 2; k++) {
        lvgpu_blend_flush_batch_t d = batch(dst[k], src[k], pan[k], n);
        CHECK(lvgpu_synth(dst[k], LVGPU_SYNTH_U16, 0x5EED0001u, 0, total, NULL));
        CHECK(lvgpu_synth(src[k], LVGPU_SYNTH_ARGB_UI, 0x5EED0002u, 0, total, NULL));
        CHECK(lvgpu_blend_flush_batch(&d, NULL));
        CHECK(lvgpu_memset(sums, 0, 2 * sizeof(uint64_t), NULL));
        CHECK(lvgpu_checksum(dst[k], 2, 0, total, sums + 0, NULL));
        CHECK(lvgpu_checksum(pan[k], 2, 0, total, sums + 1, NULL));
        uint64_t h[2];
        CHECK(lvgpu_memcpy_d2h(h, sums, sizeof(h), NULL));
        CHECK(lvgpu_stream_synchronize(NULL));
        printf("draw_buffers_%s %" PRIu64 "\npanel_images_%s %" PRIu64 "\n", name[k], h[0], name[k], h[1]);
        const double ms = time_launch(&d, 20);
        if(ms < 0.0) return 1;
        printf("%s_ms %.4f\n%s_pct_of_8TBs %.1f\n", name[k], ms, name[k], 10.0 * total / (ms * 1e-3) / 8e12 * 100.0);
    }
    printf("kernel %s\n", lvgpu_last_kernel());
    CHECK(lvgpu_free(sums));
    CHECK(lvgpu_free(arena));
    return 0;
}
