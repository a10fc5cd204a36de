// lvgpu_rows.hip -- launch of the clipped-area kernel (a translation unit of its own: see lvgpu_launch.hpp).
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>

#include "lvgpu_launch.hpp"

namespace lvgpu {
namespace {

template <int SRC>
struct LaunchRows {
    static int run(const BlendArgs & a, hipStream_t st)
    {
        RowsMap map;
        memset(&map, 0, sizeof(map));
        map.groups_per_row = ((uint32_t)a.w + 7 + 7) / 8;  // the worst row alignment
        if((a.dst_stride % 8) == 0 && (a.dst_fs % 8) == 0) {  // one alignment for every row of every frame
            const uint32_t off = (uint32_t)((reinterpret_cast<uintptr_t>(a.dst) >> 1) & 7u);
            map.groups_per_row = (off + (uint32_t)a.w + 7) / 8;
        }
        if(a.span_px) {  // whole pieces of span_px pixels around every row, and the 16-byte groups that hold them
            map.groups_per_row = ((uint32_t)a.w + 2u * ((uint32_t)a.span_px - 1u) + 7u + 7u) / 8u;
            if((a.dst_stride % a.span_px) == 0 && (a.dst_fs % a.span_px) == 0) {  // one alignment for every row of every frame
                const uint32_t A = (uint32_t)(reinterpret_cast<uintptr_t>(a.dst) >> 1), sm = (uint32_t)a.span_px - 1u;
                int lo = -(int)(A & sm), hi = a.w + (int)((0u - (A + (uint32_t)a.w)) & sm);
                lo = std::max(lo, -a.row_x0);
                hi = std::min(hi, a.dst_stride - a.row_x0);
                const int first = lo - (int)((A + (uint32_t)lo) & 7u);
                map.groups_per_row = (uint32_t)((hi - first + 7) / 8);
            }
        }
        map.src_span = (long long)(a.h - 1) * a.src_stride + a.w;
        map.mask_span = (long long)(a.h - 1) * a.mask_stride + a.w;
        const uint64_t lanes = (uint64_t)map.groups_per_row * (uint32_t)a.h;
        map.blocks_per_frame = (uint32_t)((lanes + 256 * ROWS_UNROLL - 1) / (256 * ROWS_UNROLL));
        map.frames_in_grid = (uint32_t)std::min<uint64_t>((uint64_t)a.n_frames, std::max<uint64_t>(1, 0x7FFFFFFFull / map.blocks_per_frame));
        map.remap = getenv("LVGPU_ROWS_REMAP") ? atoi(getenv("LVGPU_ROWS_REMAP")) : 0;
        dim3 grid(map.blocks_per_frame * map.frames_in_grid);
        const uint32_t rows_pad = 0;  // no residency cap: with a row's edges handled in its own wavefronts it only loses
        // the combinations resolve_blend produces with the normal blend function; everything else decides per pixel
        constexpr bool own_alpha = SRC == SRC_ARGB8888 || SRC == SRC_RGBA8888 || SRC == SRC_ARGB8565;
        const bool normal = a.blend_fn == BF_NORMAL;
        if(normal && a.opa_mode == OM_CONST && a.opa == 255u && !own_alpha && SRC != SRC_COPY)
            hipLaunchKernelGGL((k_blend_rows<SRC, OM_OPAQUE>), grid, dim3(256), rows_pad, st, a, map);
        else if(normal && a.opa_mode == OM_CONST && !own_alpha && SRC != SRC_COPY)
            hipLaunchKernelGGL((k_blend_rows<SRC, OM_CONST>), grid, dim3(256), rows_pad, st, a, map);
        else if(normal && a.opa_mode == OM_MASK && SRC != SRC_COPY)
            hipLaunchKernelGGL((k_blend_rows<SRC, OM_MASK>), grid, dim3(256), rows_pad, st, a, map);
        else if(normal && a.opa_mode == OM_FILL_MASK_OPA && SRC == SRC_FILL)
            hipLaunchKernelGGL((k_blend_rows<SRC, OM_FILL_MASK_OPA>), grid, dim3(256), rows_pad, st, a, map);
        else if(normal && a.opa_mode == OM_MAP_MASK_OPA && SRC != SRC_FILL && SRC != SRC_COPY)
            hipLaunchKernelGGL((k_blend_rows<SRC, OM_MAP_MASK_OPA>), grid, dim3(256), rows_pad, st, a, map);
        else
            hipLaunchKernelGGL((k_blend_rows<SRC, OM_RUNTIME>), grid, dim3(256), rows_pad, st, a, map);
        LVGPU_LAUNCH_TRY(hipGetLastError());
        return LVGPU_OK;
    }
};

}  // namespace

int launch_rows(int fmt, const BlendArgs & a, hipStream_t st) { return dispatch_src<LaunchRows>(fmt, a, st); }

}  // namespace lvgpu
