// lvgpu_tiles_col.hip -- launch of k_tile_col (a translation unit of its own: see lvgpu_launch.hpp).
#include <algorithm>
#include <cstring>
#include <string>

#include "lvgpu_launch.hpp"

namespace lvgpu {
namespace {
template <int SRC>
struct LaunchTileCol {
    static int run(const BlendArgs & a, hipStream_t st)
    {
        const int W = a.w, H = a.h;
        const uint32_t tiles_x = (W + TILE - 1) / TILE, cols_y = (H + TILE * COL_WPB - 1) / (TILE * COL_WPB);
        const uint64_t n_blocks = (uint64_t)tiles_x * cols_y * a.n_frames;
        if(n_blocks > 0x7FFFFFFFull) return host_fail(LVGPU_ERR_ARG, "batch too large for one launch");
        const dim3 grid((uint32_t)n_blocks), block(64 * COL_WPB);
        // Rows per phase (tools/bench_phases.py; LVGPU_COL_NR overrides).  With the blend fused in, 8 rows per phase need
        // 145 registers and a CU holds one workgroup: fine while most of its eight wavefronts have rows to work on
        // (3840x2160 64 % of peak against 62 % at 4 rows per phase, 1080p 64 against 62 %), not when a column of the
        // frame leaves many of them idle: at 4 rows per phase (88 registers) two workgroups share a CU and 480x328 goes
        // from 61 to 66 %, 480x272 from 53 to 66 %, 1024x600 from 52 to 58 %, 1280x720 from 58 to 60 %.  The rotation
        // alone gains one to three points from 4 rows per phase at every height but 4K canvases, which like 2 (63 -> 65 %).
        const int nr_chosen = SRC == SRC_COPY ? (H > 1536 ? 2 : 4) : (H <= 768 ? 4 : 8);
        with_nr_discard<SRC>(tile_rows_per_phase("LVGPU_COL_NR", nr_chosen), a.discard_dst != 0, [&](auto nr_c, auto discard_c) {
            hipLaunchKernelGGL((k_tile_col<SRC, decltype(nr_c)::value, decltype(discard_c)::value>), grid, block, 0, st, a, W, H,
                               tiles_x, cols_y, (uint32_t)n_blocks);
        });
        LVGPU_LAUNCH_TRY(hipGetLastError());
        return LVGPU_OK;
    }
};
}  // namespace

int launch_tile_col(int fmt, const BlendArgs & a, hipStream_t st) { return dispatch_src<LaunchTileCol>(fmt, a, st); }

}  // namespace lvgpu
