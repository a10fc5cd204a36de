// lvgpu_tiles.hip -- launches of the 90/270-degree tile kernels (a translation unit of its own: see lvgpu_launch.hpp).
#include <algorithm>
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
                                             : (fast_path<SRC>(a) == FAST_MIX ? residency_pad(st, TILE_PAD_LDS) : 0);
        hipLaunchKernelGGL(k_tile<SRC>, dim3((uint32_t)((n_tiles + TILE_WPB - 1) / TILE_WPB)), dim3(64 * TILE_WPB), pad, st, a, W, H,
                           tiles_x, tiles_y, tpf, (uint32_t)n_tiles);
        LVGPU_LAUNCH_TRY(hipGetLastError());
        return LVGPU_OK;
    }
};
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
        const uint32_t tiles_x = (W + TILE - 1) / TILE, cols_y = (H + TILE * ANY_WPB - 1) / (TILE * ANY_WPB);
        const uint64_t n_blocks = (uint64_t)tiles_x * cols_y * a.n_frames;
        if(n_blocks > 0x7FFFFFFFull) return host_fail(LVGPU_ERR_ARG, "batch too large for one launch");
        hipLaunchKernelGGL(k_tile_any<SRC>, dim3((uint32_t)n_blocks), dim3(64 * ANY_WPB), 0, st, a, W, H, tiles_x, cols_y, (uint32_t)n_blocks);
        LVGPU_LAUNCH_TRY(hipGetLastError());
        return LVGPU_OK;
    }
};
}  // namespace

int launch_tile(int fmt, const BlendArgs & a, hipStream_t st) { return dispatch_src<LaunchTile>(fmt, a, st); }
int launch_tile_col(int fmt, const BlendArgs & a, hipStream_t st) { return dispatch_src<LaunchTileCol>(fmt, a, st); }
int launch_tile_any(int fmt, const BlendArgs & a, hipStream_t st) { return dispatch_src<LaunchTileAny>(fmt, a, st); }

}  // namespace lvgpu
