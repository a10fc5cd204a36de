/*
 * multi_gpu_blend_flush.c -- the batched blend + flush sharded over several GPUs by a plain C host on the C ABI
 * (include/lvgpu.h only): one host thread and one stream per device, every thread calls lvgpu_init(device) for itself,
 * frames are independent units, so thread g of G takes the contiguous global frames [g*N/G, (g+1)*N/G) and the devices
 * never talk to each other.  Inputs are a pure function of the global element index (lvgpu_synth), so any partition
 * reproduces the same batch; every shard is summed with lvgpu_checksum at its global offset and the host adds the
 * per-shard sums: the totals do not depend on the number of shards.
 *
 *   multi_gpu_blend_flush [total_frames [shards [devices [rotation]]]]
 *     shards   host threads (default: one per visible device)
 *     devices  devices to spread them over, shard g runs on device g % devices (default: all visible; 1 rehearses
 *              the sharded host on a single GPU, which is what tests/test_gpu_c_host.py does)
 *
 *   gcc -O2 -pthread -Iinclude examples/multi_gpu_blend_flush.c -Lvvc_amd/lib -llvgpu -Wl,-rpath,'$ORIGIN/../vvc_amd/lib' -o examples/multi_gpu_blend_flush
 */
#include <inttypes.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "lvgpu.h"

enum { W = 480, H = 320 };

typedef struct {
    int shard, device, rotation;
    int64_t lo, hi;          /* global frame range */
    uint64_t sum_dst, sum_panel;
    double seconds;          /* the fused launch alone, host clock around launch + stream synchronisation */
    int rc;
    char err[256];
    char kernel[64];
} shard_t;

#define TRY(call)                                                                   \
    do {                                                                            \
        int rc_ = (call);                                                           \
        if(rc_ != LVGPU_OK) {                                                       \
            s->rc = rc_;                                                            \
            snprintf(s->err, sizeof(s->err), "%s: %s", #call, lvgpu_last_error()); \
            return NULL;                                                            \
        }                                                                           \
    } while(0)

static double now(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

static void * run_shard(void * arg)
{
    shard_t * s = (shard_t *)arg;
    const size_t px = (size_t)W * H, n = (size_t)(s->hi - s->lo), total = px * n;
    const uint64_t first = (uint64_t)s->lo * px; /* global element index of the shard's first pixel */
    if(n == 0) return NULL;
    TRY(lvgpu_init(s->device)); /* per thread: selects the device for this thread */
    void *stream, *dst, *src, *panel;
    uint64_t * sums;
    TRY(lvgpu_stream_create(&stream));
    TRY(lvgpu_malloc(&dst, total * 2));
    TRY(lvgpu_malloc(&src, total * 4));
    TRY(lvgpu_malloc(&panel, total * 2));
    TRY(lvgpu_malloc((void **)&sums, 2 * sizeof(uint64_t)));
    TRY(lvgpu_memset(sums, 0, 2 * sizeof(uint64_t), stream));
    TRY(lvgpu_synth(dst, LVGPU_SYNTH_U16, 0x5EED0001u, first, total, stream));
    TRY(lvgpu_synth(src, LVGPU_SYNTH_ARGB_UI, 0x5EED0002u, first, total, stream));
    TRY(lvgpu_stream_synchronize(stream));

    lvgpu_blend_flush_batch_t d;
    memset(&d, 0, sizeof(d));
    d.blend.dst = dst;
    d.blend.dst_frame_stride = (int64_t)px;
    d.blend.buf_area = (lvgpu_area_t){0, 0, W - 1, H - 1};
    d.blend.clip_area = d.blend.buf_area;
    d.blend.blend_area = d.blend.buf_area;
    d.blend.src = src;
    d.blend.src_frame_stride = (int64_t)px;
    d.blend.src_format = LVGPU_SRC_ARGB8888;
    d.blend.opa = 255;
    d.blend.color_16_swap = 1;
    d.blend.n_frames = (int32_t)n;
    d.out = panel;
    d.out_frame_stride = (int64_t)px;
    d.rotation = s->rotation;
    const double t0 = now();
    TRY(lvgpu_blend_flush_batch(&d, stream));
    snprintf(s->kernel, sizeof(s->kernel), "%s", lvgpu_last_kernel());
    TRY(lvgpu_stream_synchronize(stream));
    s->seconds = now() - t0;

    TRY(lvgpu_checksum(dst, 2, first, total, sums + 0, stream));
    TRY(lvgpu_checksum(panel, 2, first, total, sums + 1, stream));
    uint64_t h[2];
    TRY(lvgpu_memcpy_d2h(h, sums, sizeof(h), stream));
    TRY(lvgpu_stream_synchronize(stream));
    s->sum_dst = h[0];
    s->sum_panel = h[1];
    TRY(lvgpu_free(dst));
    TRY(lvgpu_free(src));
    TRY(lvgpu_free(panel));
    TRY(lvgpu_free(sums));
    TRY(lvgpu_stream_destroy(stream));
    return NULL;
}

int main(int argc, char ** argv)
{
    const int visible = lvgpu_device_count();
    if(visible <= 0) {
        fprintf(stderr, "no HIP device visible\n");
        return 1;
    }
    const int64_t total_frames = argc > 1 ? atoll(argv[1]) : 4096;
    int shards = argc > 2 ? atoi(argv[2]) : visible;
    int devices = argc > 3 ? atoi(argv[3]) : visible;
    const int rotation = argc > 4 ? atoi(argv[4]) : LVGPU_ROT_90;
    if(total_frames < 0 || shards < 1 || shards > 64 || devices < 1 || devices > visible) {
        fprintf(stderr, "bad arguments (%d device(s) visible)\n", visible);
        return 1;
    }
    shard_t s[64];
    pthread_t th[64];
    memset(s, 0, sizeof(s));
    for(int g = 0; g < shards; g++) {
        s[g].shard = g;
        s[g].device = g % devices;
        s[g].rotation = rotation;
        s[g].lo = total_frames * g / shards;
        s[g].hi = total_frames * (g + 1) / shards;
    }
    const double t0 = now();
    for(int g = 0; g < shards; g++)
        if(pthread_create(&th[g], NULL, run_shard, &s[g]) != 0) {
            fprintf(stderr, "pthread_create failed\n");
            return 1;
        }
    for(int g = 0; g < shards; g++) pthread_join(th[g], NULL);
    const double wall = now() - t0;

    uint64_t sum_dst = 0, sum_panel = 0;
    double slowest = 0;
    for(int g = 0; g < shards; g++) {
        if(s[g].rc != LVGPU_OK) {
            fprintf(stderr, "shard %d on device %d failed (%d): %s\n", g, s[g].device, s[g].rc, s[g].err);
            return 1;
        }
        sum_dst += s[g].sum_dst; /* modulo 2^64: the checksum is a sum over elements */
        sum_panel += s[g].sum_panel;
        if(s[g].seconds > slowest) slowest = s[g].seconds;
        printf("shard %d device %d frames %" PRId64 "..%" PRId64 " kernel %s launch_ms %.3f\n", g, s[g].device, s[g].lo, s[g].hi,
               s[g].kernel, s[g].seconds * 1e3);
    }
    printf("total_frames %" PRId64 "\nshards %d\ndevices %d\n", total_frames, shards, devices);
    printf("draw_buffers %" PRIu64 "\npanel_images %" PRIu64 "\n", sum_dst, sum_panel);
    if(slowest > 0)
        printf("aggregate_Mpx_s %.1f\nwall_s %.3f\n", (double)total_frames * W * H / slowest / 1e6, wall);
    return 0;
}
