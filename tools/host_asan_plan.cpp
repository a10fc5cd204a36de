// Host-side checker: the parts of the C ABI that are pure host arithmetic (display-list planning, transport chunks,
// panel transactions) driven with random input against a build of the library whose HOST code carries
// AddressSanitizer.  No GPU is needed: none of these entry points touches the device.
//   hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -Xarch_host -fsanitize=address -shared \
//         -o /tmp/asan/liblvgpu_asan.so vvc_amd/csrc/lvgpu.hip vvc_amd/csrc/lvgpu_tiles.hip \
//         vvc_amd/csrc/lvgpu_rows.hip -x hip vvc_amd/csrc/lvgpu_lvgl_shim.cpp -ldl
//   g++ -O1 -g -fsanitize=address -Iinclude tools/host_asan_plan.cpp -o /tmp/asan/plan_fuzz /tmp/asan/liblvgpu_asan.so
//   ASAN_OPTIONS=detect_leaks=0 /tmp/asan/plan_fuzz 2000
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "lvgpu.h"

int main(int argc, char ** argv)
{
    const int rounds = argc > 1 ? atoi(argv[1]) : 500;
    std::mt19937 rng(7);
    auto U = [&](int lo, int hi) { return (int)(rng() % (uint32_t)(hi - lo + 1)) + lo; };
    long long plans = 0, items_total = 0, errors = 0;
    for(int r = 0; r < rounds; r++) {
        const int W = U(1, 70) * 8 + (U(0, 9) == 0 ? U(1, 7) : 0), H = U(1, 400);
        const int n = U(0, 300);
        const bool argb = U(0, 5) == 0;
        void * const dst = (void *)(uintptr_t)(0x100000 + (U(0, 19) == 0 ? 2 : 0));  // now and then not 16-byte aligned
        std::vector<lvgpu_blend_batch_t> it((size_t)n);
        for(auto & b : it) {
            memset(&b, 0, sizeof(b));
            b.dst = dst;
            b.dst_frame_stride = (int64_t)W * H;
            b.n_frames = 3;
            b.buf_area = lvgpu_area_t{0, 0, (int16_t)(W - 1), (int16_t)(H - 1)};
            int x1 = U(-5, W + 4), x2 = U(-5, W + 4), y1 = U(-5, H + 4), y2 = U(-5, H + 4);
            if(U(0, 3) == 0) y2 = y1;  // one-row items, as LVGL's producers issue them
            if(x2 < x1) std::swap(x1, x2);
            if(y2 < y1) std::swap(y1, y2);
            b.blend_area = lvgpu_area_t{(int16_t)x1, (int16_t)y1, (int16_t)x2, (int16_t)y2};
            b.mask_area = b.blend_area;
            b.clip_area = U(0, 2) ? b.buf_area : lvgpu_area_t{(int16_t)U(0, W - 1), (int16_t)U(0, H - 1), (int16_t)(W - 1), (int16_t)(H - 1)};
            const int kind = U(0, 5);
            b.src_format = kind <= 2 ? LVGPU_SRC_FILL : (kind == 3 ? LVGPU_SRC_RGB565 : (kind == 4 ? LVGPU_SRC_ARGB8888 : LVGPU_SRC_ARGB8565));
            if(b.src_format != LVGPU_SRC_FILL) b.src = (const void *)(uintptr_t)0x200000;
            if(b.src_format <= LVGPU_SRC_RGB565 && U(0, 1)) b.mask = (const uint8_t *)(uintptr_t)0x300000;
            b.mask_res = LVGPU_MASK_RES_CHANGED;
            b.color = (uint16_t)rng();
            const int opas[] = {0, 2, 3, 100, 135, 252, 253, 255};
            b.opa = (uint8_t)opas[U(0, 7)];
            b.blend_mode = (uint8_t)(U(0, 7) == 0 ? U(1, 3) : 0);
            b.color_16_swap = 1;
            b.dst_argb8565 = argb ? 1 : 0;
        }
        lvgpu_list_plan_t p;
        const int flush = argb ? 0 : U(0, 1), rot = U(0, 3);
        const int rc = lvgpu_blend_list_plan(n ? &it[0] : nullptr, it.data(), n, flush, rot, &p);
        if(rc != LVGPU_OK) {
            errors++;  // rejected as a whole
            continue;
        }
        plans++;
        items_total += p.items_listed + p.items_standalone + p.items_noop;
        if(p.items_listed + p.items_standalone + p.items_noop > n) {
            fprintf(stderr, "plan counts more items than given\n");
            return 1;
        }
    }
    long long chunks = 0;
    for(int r = 0; r < rounds; r++) {
        const int W = U(1, 500), H = U(1, 500), rot = U(0, 3);
        const uint32_t trans = (uint32_t)U(1, W * H + 10);
        const lvgpu_area_t a{0, 0, (int16_t)(W - 1), (int16_t)(H - 1)};
        const int n = lvgpu_flush_chunks(&a, W, H, rot, trans, nullptr, 0);
        if(n <= 0) continue;
        std::vector<lvgpu_flush_chunk_t> c((size_t)n);
        if(lvgpu_flush_chunks(&a, W, H, rot, trans, c.data(), n) != n) return 2;
        long long px = 0;
        for(auto & k : c) {
            px += k.n_px;
            lvgpu_panel_txn_t t[3];
            if(lvgpu_axs15231b_transactions(&k, U(0, 3), U(0, 3), U(0, 1), 16, t) < 2) return 3;
        }
        if(px != (long long)W * H) {
            fprintf(stderr, "chunks do not cover the area: %lld of %d\n", px, W * H);
            return 4;
        }
        chunks += n;
    }
    printf("host checker ok: %lld plans (%lld items, %lld lists rejected as a whole), %lld transport chunks\n", plans, items_total, errors, chunks);
    return 0;
}
