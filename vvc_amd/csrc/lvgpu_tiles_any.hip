// lvgpu_tiles_any.hip -- launch of k_tile_any (a translation unit of its own: see lvgpu_launch.hpp).
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>

#include "lvgpu_launch.hpp"

namespace lvgpu {
namespace {
template <int SRC>
struct LaunchTileAny {
    static int run(const BlendArgs & a, hipStream_t st)
    {
        const int W = a.w, H = a.h;
        // Rows per phase and wavefronts per workgroup (measured with tools/bench_phases.py; LVGPU_ANY_NR and
        // LVGPU_ANY_WPB in the environment override them).  A workgroup takes 64 * wpb rows of a 64-pixel column of the
        // frame; wavefronts with no rows left idle at its barrier and hold registers and LDS meanwhile, so among the
        // candidates the count that launches the fewest wavefronts over a column wins, the order deciding ties.
        //   The rotation alone (46 registers at 8 rows per phase): one workgroup per column of 320 or 384 rows
        // (482x320: 67 % of the HBM peak with 5 wavefronts, 52 % with 4; 480x322: 64 % with 6, 50 % with 3), not above
        // six (466x466: 61 % with 4, 57 % with 8); frames above 400 rows gain from 2 rows per phase (466x466 60 -> 62 %,
        // 1080p 64 -> 68 %), lower ones lose (135x240 66 -> 62 %).
        //   With the blend fused in, 8 rows per phase need 133 registers: three wavefronts per SIMD, and a workgroup
        // above four wavefronts has a CU to itself (240x135: 59 % with 3, 30 % with 5 to 8).  At 2 rows per phase (61
        // registers) the workgroup can be sized to the height like the rotation's: 466x466 52 -> 60 % (8 wavefronts),
        // 481x321 49 -> 60 %, 480x322 53 -> 65 %, 482x320 58 -> 62 %, 390x390 50 -> 59 %.  Frames of up to four tiles
        // of height keep 8 rows per phase (320x172 67 against 66 %, 320x200 68 against 65 %, 240x135 and 240x240 level).
        const int tiles_y = (H + TILE - 1) / TILE;
        //   From 9 to 16 tiles of height (513 to 1024 rows) one workgroup takes a column of the frame whole, as the column
        // kernel does: 1024x600 fused 57 -> 66 % (10 wavefronts, 4 rows per phase), rotation alone 62 -> 69 %; 1280x720
        // fused 59 -> 65 %.  Such workgroups cannot have the whole block per phase (128 registers per lane).
        // The rotation alone only while two such workgroups fit a CU's LDS (up to 10 wavefronts: 1023x600 55 -> 65 %;
        // with 12, one workgroup per CU, 1278x718 falls from 61 to 53 % and 722x721 from 60 to 49 %).
        const bool whole_column = tiles_y >= 9 && tiles_y <= (SRC == SRC_COPY ? 10 : ANY_WPB_MAX);
        const int nr = tile_rows_per_phase("LVGPU_ANY_NR", SRC == SRC_COPY ? (H > 400 ? 2 : 8) : (tiles_y <= 4 ? 8 : (whole_column ? 4 : 2)));
        static const int order_copy[6] = {6, 5, 4, 3, 7, 8}, order_blend8[2] = {4, 3}, order_blend[5] = {6, 8, 7, 5, 4};
        const int * order = SRC == SRC_COPY ? order_copy : (nr == 8 ? order_blend8 : order_blend);
        const int n_order = SRC == SRC_COPY ? 6 : (nr == 8 ? 2 : 5);
        int wpb = tiles_y;
        if(wpb > 2) {
            uint32_t slots = 0xFFFFFFFFu;
            for(int i = 0; i < n_order; i++) {
                const uint32_t s = (uint32_t)((H + TILE * order[i] - 1) / (TILE * order[i])) * order[i];
                if(s < slots) { slots = s; wpb = order[i]; }
            }
            // five tiles of height, blend fused in: six wavefronts (three workgroups per CU) are ahead of five (four per CU)
            if(SRC != SRC_COPY && nr < 8 && tiles_y == 5) wpb = 6;
            if(whole_column && nr < 8) wpb = tiles_y;
        }
        wpb = std::max(1, std::min(wpb, (int)(max_lds_per_block() / 8192)));   // 8 KB of LDS per wavefront
        if(const char * e = getenv("LVGPU_ANY_WPB")) wpb = std::min(std::max(atoi(e), 1), any_wpb_max(nr));
        const uint32_t cols_y = (H + TILE * wpb - 1) / (TILE * wpb);
        const uint32_t tiles_x = (W + TILE - 1) / TILE;
        const uint64_t n_blocks = (uint64_t)tiles_x * cols_y * a.n_frames;
        if(n_blocks > 0x7FFFFFFFull) return host_fail(LVGPU_ERR_ARG, "batch too large for one launch");
        const dim3 grid((uint32_t)n_blocks), block(64 * wpb);
        const size_t lds = (size_t)wpb * 8192;
        with_nr_discard<SRC>(nr, a.discard_dst != 0, [&](auto nr_c, auto discard_c) {
            hipLaunchKernelGGL((k_tile_any<SRC, decltype(nr_c)::value, decltype(discard_c)::value>), grid, block, lds, st, a, W, H,
                               tiles_x, cols_y, wpb);
        });
        LVGPU_LAUNCH_TRY(hipGetLastError());
        return LVGPU_OK;
    }
};
}  // namespace

int launch_tile_any(int fmt, const BlendArgs & a, hipStream_t st) { return dispatch_src<LaunchTileAny>(fmt, a, st); }

}  // namespace lvgpu
