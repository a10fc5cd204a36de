// lvgpu_tiles_col.hip -- launch of k_tile_col (a translation unit of its own: see lvgpu_launch.hpp).
#include <algorithm>
#include <cstdlib>
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
        // Wavefronts per workgroup and rows per phase (tools/exp/r3_col_wpb.py; LVGPU_COL_WPB and LVGPU_COL_NR override).
        // A workgroup takes 64 * wpb rows of a 64-pixel column of the frame.  Up to 1024 rows one workgroup as high as
        // the frame writes whole panel rows, which follow each other in memory: with eight-wavefront workgroups a
        // column of 600 rows is two workgroups, six of whose sixteen wavefronts have nothing to load, and every panel row
        // is written as two runs -- 1024x600 fused 58.6 -> 67.9 % of the HBM peak, 1280x720 61.4 -> 68.2 %, 720x720
        // 55.9 -> 65.2 %; the rotation alone 54.9 -> 70.7 %, 59.1 -> 70.2 %, 55.6 -> 65.4 %.  Above 1024 rows: the count
        // from 9 to 16 that leaves the fewest idle wavefronts over a column (1080p: 9, fused 63.9 -> 64.8 %), and 16 on
        // the highest frames, where the longer runs count for more than the idle wavefronts (3840x2160: rotation alone
        // 64.6 -> 66.9 %, without the write-back 66.0 -> 67.7 %, fused 64.8 -> 64.9 %).
        //   Rows per phase: workgroups above eight wavefronts cannot have the whole block per phase (its blend needs more
        // than their 128 registers per lane) and nothing is lost by that: 4 rows per phase (88 registers) are level with
        // or ahead of 8 wherever both can run (1080p 64.8 against 63.9 %, 4K level); without the write-back 2 rows per
        // phase are ahead by up to a point.  With the write-back, frames of up to 512 rows keep eight wavefronts, two
        // workgroups to a CU (480x272: 67.5 against 65.1 % with five); without it a workgroup as high as the frame is ahead
        // there too (480x272 65.1 -> 70.0 %, 360x360 66.1 -> 68.7 %).
        const int tiles_y = (H + TILE - 1) / TILE;
        auto fewest_idle = [tiles_y]() {
            int best = COL_WPB_MAX, slots = 1 << 30;
            for(int w = 9; w <= COL_WPB_MAX; w++) {
                const int s = (tiles_y + w - 1) / w * w;
                if(s <= slots) { slots = s; best = w; }
            }
            return best;
        };
        const int wpb_tall = tiles_y <= COL_WPB_MAX ? tiles_y : (tiles_y > 24 ? COL_WPB_MAX : fewest_idle());
        const bool two_tiles = tiles_y > COL_WPB_MAX && (size_t)tiles_y * 8192 <= max_lds_per_block();   // see tpw below
        const int nr_chosen = SRC == SRC_COPY ? 4 : (a.discard_dst && !two_tiles ? 2 : 4);
        const int nr = tile_rows_per_phase("LVGPU_COL_NR", nr_chosen);
        // (up to three tiles of height four wavefronts do as well as eight and leave fewer idle: 80x160 51 -> 58 %)
        int wpb = (SRC != SRC_COPY && !a.discard_dst && tiles_y <= 8) ? (tiles_y <= 3 ? 4 : 8) : wpb_tall;
        wpb = std::min(wpb, col_wpb_max(nr));
        wpb = std::max(1, std::min(wpb, (int)(max_lds_per_block() / 8192)));   // 8 KB of LDS per wavefront
        if(const char * e = getenv("LVGPU_COL_WPB")) wpb = std::min(std::max(atoi(e), 1), col_wpb_max(nr));
        // Frames of 17 to 20 tiles of height (1080p, 1600x1200): still one workgroup per column, its wavefronts taking two
        // tiles each, one after the other, where the device's LDS holds the column (8 KB per tile, 136 to 160 KB):
        // 1080p fused 64.8 -> 67.6 %, rotation alone 64.2 -> 69.5 %, without the write-back 65.9 -> 69.7 %; 1600x1200
        // 66.0 -> 67.4 %, 65.3 -> 69.5 %, 66.7 -> 68.9 %.  Higher frames, which need more than one workgroup per column
        // either way, gain nothing from it (3840x2160, 2560x1440), lower ones lose (1280x720 68.1 -> 66.4 %).
        int tpw = 1;
        if(tiles_y > COL_WPB_MAX && (size_t)tiles_y * 8192 <= max_lds_per_block() && nr < 8 && !getenv("LVGPU_COL_WPB")) {
            tpw = 2;
            wpb = (tiles_y + 1) / 2;
        }
        if(const char * e = getenv("LVGPU_COL_TPW")) tpw = atoi(e) == 2 ? 2 : (atoi(e) == 4 ? 4 : 1);   // for measuring
        if((size_t)wpb * tpw * 8192 > max_lds_per_block()) tpw = 1;
        const uint32_t tiles_x = (W + TILE - 1) / TILE, cols_y = (H + TILE * wpb * tpw - 1) / (TILE * wpb * tpw);
        const uint64_t n_blocks = (uint64_t)tiles_x * cols_y * a.n_frames;
        if(n_blocks > 0x7FFFFFFFull) return host_fail(LVGPU_ERR_ARG, "batch too large for one launch");
        const dim3 grid((uint32_t)n_blocks), block(64 * wpb);
        const size_t lds = (size_t)wpb * tpw * 8192;
        with_nr_discard<SRC>(nr, a.discard_dst != 0, [&](auto nr_c, auto discard_c) {
            hipLaunchKernelGGL((k_tile_col<SRC, decltype(nr_c)::value, decltype(discard_c)::value>), grid, block, lds, st, a, W, H,
                               tiles_x, cols_y, wpb, tpw);
        });
        LVGPU_LAUNCH_TRY(hipGetLastError());
        return LVGPU_OK;
    }
};
}  // namespace

int launch_tile_col(int fmt, const BlendArgs & a, hipStream_t st) { return dispatch_src<LaunchTileCol>(fmt, a, st); }

}  // namespace lvgpu
