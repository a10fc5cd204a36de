// lvgpu_rows.hip -- launch of the clipped-area kernel (a translation unit of its own: see lvgpu_launch.hpp).
#include <algorithm>
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
        map.src_span = (long long)(a.h - 1) * a.src_stride + a.w;
        map.mask_span = (long long)(a.h - 1) * a.mask_stride + a.w;
        const uint64_t lanes = (uint64_t)map.groups_per_row * (uint32_t)a.h;
        dim3 grid((uint32_t)((lanes + 256 * ROWS_UNROLL - 1) / (256 * ROWS_UNROLL)), std::min<uint32_t>(a.n_frames, 65535));
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

template <int SRC>
struct LaunchArgbRows {
    static int run(const BlendArgs & a, hipStream_t st)
    {
        RowsMap map;
        memset(&map, 0, sizeof(map));
        map.groups_per_row = (((uint32_t)a.w + 7) / 8 + 1) & ~1u;   // lanes per row: even, see the kernel
        const uint64_t lanes = (uint64_t)map.groups_per_row * (uint32_t)a.h;
        // one frame per lane: lanes that walk several frames are slower (468x296 area: 65.2 % of the HBM peak with one,
        // 63.8 / 62.8 / 61.1 % with 2 / 4 / 16 -- a lane's loads of the next frame wait for its stores of this one)
        dim3 grid((uint32_t)((lanes + 255) / 256), std::min<uint32_t>(a.n_frames, 65535));
        hipLaunchKernelGGL(k_blend_argb_rows<SRC>, grid, dim3(256), 0, st, a, map);
        LVGPU_LAUNCH_TRY(hipGetLastError());
        return LVGPU_OK;
    }
};

}  // namespace

int launch_rows(int fmt, const BlendArgs & a, hipStream_t st) { return dispatch_src<LaunchRows>(fmt, a, st); }
int launch_argb_rows(int fmt, const BlendArgs & a, hipStream_t st) { return dispatch_src<LaunchArgbRows>(fmt, a, st); }

}  // namespace lvgpu
