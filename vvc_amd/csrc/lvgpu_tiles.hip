// lvgpu_tiles.hip -- launches of the 90/270-degree tile kernels (a translation unit of its own: see lvgpu_launch.hpp).
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>

#include "lvgpu_launch.hpp"

namespace lvgpu {
namespace {

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
        hipLaunchKernelGGL(k_tile<SRC>, dim3((uint32_t)((n_tiles + TILE_WPB - 1) / TILE_WPB)), dim3(64 * TILE_WPB), pad, st, a, W, H,
                           tiles_x, tiles_y, tpf, (uint32_t)n_tiles);
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
    // one workgroup per CU (RGB565 + A8 image: 68.0 % of peak uncapped, 69.3 % capped); the callers send the plain-mix case only
    const uint32_t pad = residency_pad(st, BAND_PAD_LDS);
    hipLaunchKernelGGL(k_band<SRC>, dim3((uint32_t)n_blocks), dim3(64 * BAND_WAVES), pad, st, a, W, H, bands);
    hipError_t e = hipGetLastError();
    if(e != hipSuccess && pad != 0) {
        // the padded request (64 KB used + the pad) is above what a runtime that keeps workgroups to 64 KB of LDS accepts:
        // the cap is a tuning, the launch is not
        hipLaunchKernelGGL(k_band<SRC>, dim3((uint32_t)n_blocks), dim3(64 * BAND_WAVES), 0, st, a, W, H, bands);
        e = hipGetLastError();
    }
    LVGPU_LAUNCH_TRY(e);
    return LVGPU_OK;
}
template <int SRC>
struct LaunchTileCol {
    static int run(const BlendArgs & a, hipStream_t st)
    {
        const int W = a.w, H = a.h;
        const uint32_t tiles_x = (W + TILE - 1) / TILE, cols_y = (H + TILE * COL_WPB - 1) / (TILE * COL_WPB);
        const uint64_t n_blocks = (uint64_t)tiles_x * cols_y * a.n_frames;
        if(n_blocks > 0x7FFFFFFFull) return host_fail(LVGPU_ERR_ARG, "batch too large for one launch");
        hipLaunchKernelGGL(k_tile_col<SRC>, dim3((uint32_t)n_blocks), dim3(64 * COL_WPB), 0, st, a, W, H, tiles_x, cols_y, (uint32_t)n_blocks);
        LVGPU_LAUNCH_TRY(hipGetLastError());
        return LVGPU_OK;
    }
};
template <int SRC>
struct LaunchTileAny {
    static int run(const BlendArgs & a, hipStream_t st)
    {
        const int W = a.w, H = a.h;
        // Wavefronts per workgroup: among the candidates, the count that launches the fewest wavefronts over a column
        // of the frame (idle ones hold registers and LDS until the workgroup's barrier); the order below decides ties.
        // With the blend fused in, workgroups above four wavefronts lose whatever the height (240x135: 58 % of peak
        // with 3 or 4, 32 % with 5 to 8); the rotation alone gains from a workgroup that takes a 320- or 384-row
        // column whole (482x320: 67 % with 5, 52 % with 4; 480x322: 64 % with 6, 50 % with 3) and loses above six
        // (466x466: 61 % with 4, 57 % with 8).  One or two wavefronts (128- and 256-byte runs) only where the frame is
        // no taller.  LVGPU_ANY_WPB in the environment overrides the choice (tools/bench_any_shapes.sh sweeps it).
        static const int order_copy[6] = {6, 5, 4, 3, 7, 8}, order_blend[2] = {4, 3};
        const int * order = SRC == SRC_COPY ? order_copy : order_blend;
        int wpb = (H + TILE - 1) / TILE;
        if(wpb > 2) {
            uint32_t slots = 0xFFFFFFFFu;
            for(int i = 0; i < (SRC == SRC_COPY ? 6 : 2); i++) {
                const uint32_t s = (uint32_t)((H + TILE * order[i] - 1) / (TILE * order[i])) * order[i];
                if(s < slots) { slots = s; wpb = order[i]; }
            }
        }
        if(const char * e = getenv("LVGPU_ANY_WPB")) wpb = std::min(std::max(atoi(e), 1), SRC == SRC_COPY ? ANY_WPB_MAX : ANY_WPB_MAX_BLEND);
        const uint32_t cols_y = (H + TILE * wpb - 1) / (TILE * wpb);
        const uint32_t tiles_x = (W + TILE - 1) / TILE;
        const uint64_t n_blocks = (uint64_t)tiles_x * cols_y * a.n_frames;
        if(n_blocks > 0x7FFFFFFFull) return host_fail(LVGPU_ERR_ARG, "batch too large for one launch");
        hipLaunchKernelGGL(k_tile_any<SRC>, dim3((uint32_t)n_blocks), dim3(64 * wpb), (size_t)wpb * 8192, st, a, W, H, tiles_x, cols_y, wpb);
        LVGPU_LAUNCH_TRY(hipGetLastError());
        return LVGPU_OK;
    }
};
}  // namespace

int launch_tile(int fmt, const BlendArgs & a, hipStream_t st) { return dispatch_src<LaunchTile>(fmt, a, st); }
int launch_tile_col(int fmt, const BlendArgs & a, hipStream_t st) { return dispatch_src<LaunchTileCol>(fmt, a, st); }
int launch_band(int fmt, const BlendArgs & a, hipStream_t st)
{
    switch(fmt) {
        case SRC_FILL: return launch_band_src<SRC_FILL>(a, st);
        default: return launch_band_src<SRC_RGB565>(a, st);
    }
}
int launch_tile_any(int fmt, const BlendArgs & a, hipStream_t st) { return dispatch_src<LaunchTileAny>(fmt, a, st); }

}  // namespace lvgpu
