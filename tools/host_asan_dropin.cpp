// Host-side checker for the LVGL plug-in: the real LVGL (oracle/_ref, built from the reference sources) renders widget
// scenes through lvgpu_blend / lvgpu_flush_cb, immediate and deferred, with the HOST code of the library built under
// AddressSanitizer (queue arena, staging, hooks, chunk emission).  Needs a GPU; kernels are not instrumented.
//   hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -Xarch_host -fsanitize=address -shared \
//         -o build/asan/liblvgpu_asan.so vvc_amd/csrc/lvgpu.hip vvc_amd/csrc/lvgpu_tiles.hip vvc_amd/csrc/lvgpu_tiles_col.hip \
//         vvc_amd/csrc/lvgpu_tiles_any.hip vvc_amd/csrc/lvgpu_rows.hip -x hip vvc_amd/csrc/lvgpu_lvgl_shim.cpp -ldl
//   clang++ -O1 -g -fsanitize=address -Iinclude tools/host_asan_dropin.cpp -o build/asan/dropin_check \
//         build/asan/liblvgpu_asan.so oracle/_ref/liblvgl_swap1.so -Wl,-rpath,'$ORIGIN' -Wl,-rpath,'$ORIGIN/../../oracle/_ref'
//   ASAN_OPTIONS=detect_leaks=0 build/asan/dropin_check
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <unistd.h>

#include "lvgpu.h"
#include "lvgpu_lvgl.h"

extern "C" {
// LVGL proper
void lv_draw_sw_init_ctx(void *, void *);
void lv_draw_sw_blend_basic(void *, const void *);
void lv_disp_flush_ready(void *);
// oracle/ref_harness.c
void ref_scene_set_pre_flush(void * fn);
void ref_scene_set_buf_allocator(void * alloc_fn, void * free_fn);
int ref_scene_display(int hor_res, int ver_res, void * init, void * deinit, size_t ctx_size, void * flush, void * user_data,
                      int full_refresh, int flags);
void * ref_scene_fb(void);
void * ref_scene_draw_buf(void);
int ref_scene_px_bytes(void);
int ref_scene_build(int id);
void ref_scene_refresh(void);
}

static long long g_sunk = 0;
static void sink(void *, int xs, int ys, int xe, int ye, const void * px)
{
    const uint16_t * p = (const uint16_t *)px;
    long long n = (long long)(xe - xs) * (ye - ys), s = 0;
    for(long long i = 0; i < n; i++) s += p[i];  // read every pixel handed over
    g_sunk += n + (s & 1);
}

int main()
{
    if(lvgpu_init(0) != LVGPU_OK) {
        fprintf(stderr, "lvgpu_init: %s\n", lvgpu_last_error());
        return 1;
    }
    lvgpu_lvgl_bind((void *)lv_draw_sw_init_ctx, (void *)lv_draw_sw_blend_basic, (void *)lv_disp_flush_ready);
    lvgpu_lvgl_set_color_16_swap(1);
    const int sizes[][2] = {{480, 320}, {320, 480}, {200, 96}};
    int checked = 0;
    for(int deferred = 0; deferred <= 2; deferred++) {   // 2: deferred with the draw buffer in pinned memory (drawn in place)
        lvgpu_lvgl_set_deferred(deferred ? 1 : 0);
        if(deferred == 2) ref_scene_set_buf_allocator((void *)lvgpu_lvgl_buf_alloc, (void *)lvgpu_lvgl_buf_free);
        else ref_scene_set_buf_allocator(nullptr, nullptr);
        for(int scene = 0; scene <= 3; scene++) {
            for(int flags = 0; flags <= 2; flags++) {
                for(auto & wh : sizes) {
                    const int w = wh[0], h = wh[1];
                    ref_scene_set_pre_flush(nullptr);
                    if(ref_scene_display(w, h, nullptr, nullptr, 0, nullptr, nullptr, 1, flags) != 0) return 2;
                    ref_scene_build(scene);
                    ref_scene_refresh();
                    const size_t bytes = (size_t)w * h * ref_scene_px_bytes();
                    std::vector<uint8_t> sw((uint8_t *)ref_scene_fb(), (uint8_t *)ref_scene_fb() + bytes);
                    // blend plug-in, the harness's own flush callback
                    ref_scene_set_pre_flush(deferred ? (void *)lvgpu_wait_for_finish : nullptr);
                    if(ref_scene_display(w, h, (void *)lvgpu_draw_ctx_init, (void *)lvgpu_draw_ctx_deinit, lvgpu_draw_ctx_size(),
                                         nullptr, nullptr, 1, flags) != 0)
                        return 3;
                    for(int rep = 0; rep < 4; rep++) {  // deferred mode: plan, compile + replay, replay, replay
                        ref_scene_build(scene);
                        ref_scene_refresh();
                        if(memcmp(ref_scene_fb(), sw.data(), bytes) != 0) {
                            fprintf(stderr, "frame differs: deferred %d scene %d flags %d %dx%d refresh %d\n", deferred, scene, flags, w,
                                    h, rep);
                            return 4;
                        }
                        checked++;
                    }
                    if(flags & 2) continue;  // the flush callback takes lv_color_t buffers only
                    // both plug-in points
                    for(int rot = 0; rot <= 3; rot++) {
                        lvgpu_flush_cfg_t cfg;
                        memset(&cfg, 0, sizeof(cfg));
                        cfg.sw_rotate = rot;
                        cfg.trans_size = (uint32_t)(w * h / 10);
                        cfg.draw_bitmap = sink;
                        ref_scene_set_pre_flush(nullptr);
                        if(ref_scene_display(w, h, (void *)lvgpu_draw_ctx_init, (void *)lvgpu_draw_ctx_deinit, lvgpu_draw_ctx_size(),
                                             (void *)lvgpu_flush_cb, &cfg, 1, flags) != 0)
                            return 5;
                        for(int rep = 0; rep < 3; rep++) {
                            ref_scene_build(scene);
                            ref_scene_refresh();
                            if(memcmp(ref_scene_draw_buf(), sw.data(), bytes) != 0) {
                                fprintf(stderr, "draw buffer differs: deferred %d scene %d flags %d rot %d refresh %d\n", deferred, scene,
                                        flags, rot, rep);
                                return 6;
                            }
                        }
                        ref_scene_display(16, 16, nullptr, nullptr, 0, nullptr, nullptr, 1, 0);  // cfg goes out of scope
                        checked++;
                    }
                }
            }
        }
    }
    ref_scene_set_buf_allocator(nullptr, nullptr);
    ref_scene_display(16, 16, nullptr, nullptr, 0, nullptr, nullptr, 1, 0);
    lvgpu_lvgl_stats_t st;
    lvgpu_lvgl_get_stats(&st);
    printf("drop-in checker ok: %d renders identical to LVGL's software renderer, %llu blends on the GPU, %llu lists (%llu replayed, "
           "%llu in place), %llu flushes, %lld pixels handed to the sink\n",
           checked, (unsigned long long)st.blend_gpu, (unsigned long long)st.blend_lists, (unsigned long long)st.list_replays,
           (unsigned long long)st.lists_in_place,
           (unsigned long long)st.flush_gpu, g_sunk);
    fflush(stdout);
    _exit(0);  // the sanitizer runtime of this toolchain trips over the HIP runtime's own unload order at exit
}
