// lvgpu_tiles.hip -- launches of the 90/270-degree tile and band kernels (a translation unit of its own: see
// lvgpu_launch.hpp).
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>

#include "lvgpu_launch.hpp"

namespace lvgpu {

// Rows per phase of the quarter-turn kernels (see put_transposed): the launcher's choice, or what the environment
// variable `name` says (8, 4 or 2; read at every launch -- a switch for measuring, results do not depend on it).
int tile_rows_per_phase(const char * name, int chosen)
{
    if(const char * e = getenv(name)) {
        const int v = atoi(e);
        if(v == 8 || v == 4 || v == 2) return v;
    }
    return chosen;
}

// The most LDS one workgroup may ask for on the current device (static + dynamic), looked up once per device.
// LVGPU_MAX_LDS_PER_BLOCK in the environment (bytes; read at every call) stands in for it: a switch for tests that
// the band kernel's residency pad is dropped where it does not fit.
size_t max_lds_per_block()
{
    if(const char * e = getenv("LVGPU_MAX_LDS_PER_BLOCK")) return (size_t)atoll(e);
    static thread_local int dev_cached = -1;
    static thread_local size_t cached = 0;
    int dev = -1, v = 0;
    if(hipGetDevice(&dev) != hipSuccess) return 64 * 1024;
    if(dev != dev_cached) {
        if(hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess || v <= 0) v = 64 * 1024;
        cached = (size_t)v;
        dev_cached = dev;
    }
    return cached;
}

namespace {

thread_local uint32_t g_band_pad_last = 0;  // what the last k_band launch of this thread asked for beyond its static LDS

template <int SRC>
struct LaunchTile {
    static int run(const BlendArgs & a, hipStream_t st)
    {
        const int W = a.w, H = a.h;
        const uint32_t tiles_x = (W + TILE - 1) / TILE, tiles_y = (H + TILE - 1) / TILE;
        const uint32_t tpf = tiles_x * tiles_y;
        const uint64_t n_tiles = (uint64_t)tpf * a.n_frames;
        if(n_tiles > 0xFFFFFFF0ull) return host_fail(LVGPU_ERR_ARG, "batch too large for one launch");
        const uint32_t pad = SRC == SRC_COPY ? residency_pad(st, TILE_PAD_LDS_COPY)
                                             : (fast_is_mix(fast_path<SRC>(a)) ? residency_pad(st, TILE_PAD_LDS) : 0);
        const dim3 grid((uint32_t)((n_tiles + TILE_WPB - 1) / TILE_WPB)), block(64 * TILE_WPB);
        with_nr_discard<SRC>(tile_rows_per_phase("LVGPU_TILE_NR", 8), a.discard_dst != 0, [&](auto nr_c, auto discard_c) {
            hipLaunchKernelGGL((k_tile<SRC, decltype(nr_c)::value, decltype(discard_c)::value>), grid, block, pad, st, a, W, H,
                               tiles_x, tiles_y, tpf, (uint32_t)n_tiles);
        });
        LVGPU_LAUNCH_TRY(hipGetLastError());
        return LVGPU_OK;
    }
};
template <int SRC>
int launch_band_src(const BlendArgs & a, hipStream_t st)
{
    const int W = a.w, H = a.h;
    const uint32_t bands = (H + BAND_H - 1) / BAND_H;
    const uint64_t n_blocks = (uint64_t)bands * a.n_frames;
    if(n_blocks > 0x7FFFFFFFull) return host_fail(LVGPU_ERR_ARG, "batch too large for one launch");
    // one workgroup per CU (RGB565 + A8 image: 68.0 % of peak uncapped, 69.3 % capped); the callers send the plain-mix
    // case only.  The kernel uses 64 KB of static LDS: the pad is a tuning and is left out where the device (or a
    // runtime that keeps workgroups to 64 KB) would refuse the sum -- the launch itself never depends on it.
    constexpr size_t BAND_STATIC_LDS = (size_t)BAND_H * (BAND_MAX_W / 8) * sizeof(uint4);
    uint32_t pad = residency_pad(st, BAND_PAD_LDS);
    if(BAND_STATIC_LDS + pad > max_lds_per_block()) pad = 0;
    g_band_pad_last = pad;
    if(a.discard_dst) hipLaunchKernelGGL((k_band<SRC, true>), dim3((uint32_t)n_blocks), dim3(64 * BAND_WAVES), pad, st, a, W, H, bands);
    else hipLaunchKernelGGL((k_band<SRC, false>), dim3((uint32_t)n_blocks), dim3(64 * BAND_WAVES), pad, st, a, W, H, bands);
    LVGPU_LAUNCH_TRY(hipGetLastError());
    return LVGPU_OK;
}
}  // namespace

int launch_tile(int fmt, const BlendArgs & a, hipStream_t st) { return dispatch_src<LaunchTile>(fmt, a, st); }
uint32_t band_pad_of_last_launch() { return g_band_pad_last; }
int launch_band(int fmt, const BlendArgs & a, hipStream_t st)
{
    switch(fmt) {
        case SRC_FILL: return launch_band_src<SRC_FILL>(a, st);
        default: return launch_band_src<SRC_RGB565>(a, st);
    }
}

}  // namespace lvgpu
