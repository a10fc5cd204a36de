// lvgpu_launch.hpp -- what the translation units of liblvgpu.so share on the host side.  The kernels are spread
// over several of them so that they compile side by side: lvgpu.hip (C ABI, argument resolution, the linear, ARGB8565,
// display-list and conversion kernels), lvgpu_tiles.hip (k_tile, k_band), lvgpu_tiles_col.hip (k_tile_col),
// lvgpu_tiles_any.hip (k_tile_any) and lvgpu_rows.hip (k_blend_rows).
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "../../include/lvgpu.h"
#include "lvgpu_kernels.hpp"

namespace lvgpu {

// records the calling thread's error text (lvgpu_last_error) and returns code
int host_fail(int code, const char * text);

// Dynamic LDS to ask for so that fewer workgroups share a CU (0 on a stream confined to some of the CUs, or with
// LVGPU_NO_RESIDENCY_CAP in the environment); the values sit on measured plateaus, see their uses.
constexpr uint32_t LIN_PAD_LDS = 48 * 1024, LIN_PAD_LDS_FILL = 32 * 1024, TILE_PAD_LDS = 32 * 1024, TILE_PAD_LDS_COPY = 20 * 1024,
                   BAND_PAD_LDS = 40 * 1024, ARGB_PAD_LDS = 16 * 1024, LIN_PAD_LDS_PREMULT = 24 * 1024;
uint32_t residency_pad(hipStream_t st, uint32_t pad_bytes);
int tile_rows_per_phase(const char * env_name, int chosen);
size_t max_lds_per_block();   // the most LDS (static + dynamic) one workgroup may ask for on the current device

// One launch each; fmt is a SrcFmt (SRC_COPY: rotation alone).  Shapes are checked by the callers (tile_ok, ...).
int launch_tile(int fmt, const BlendArgs & a, hipStream_t st);
int launch_tile_col(int fmt, const BlendArgs & a, hipStream_t st);
int launch_band(int fmt, const BlendArgs & a, hipStream_t st);   // fmt: SRC_FILL or SRC_RGB565
uint32_t band_pad_of_last_launch();  // dynamic LDS the calling thread's last k_band launch asked for (diagnostics)
int launch_tile_any(int fmt, const BlendArgs & a, hipStream_t st);
int launch_rows(int fmt, const BlendArgs & a, hipStream_t st);
int launch_argb_rows(int fmt, const BlendArgs & a, hipStream_t st);   // ARGB8565 destination, any rectangle

#define LVGPU_LAUNCH_TRY(expr)                                                                                      \
    do {                                                                                                            \
        hipError_t e_ = (expr);                                                                                     \
        if(e_ != hipSuccess) {                                                                                      \
            (void)hipGetLastError(); /* not to be reported again by the check after the next launch */                  \
            return host_fail(LVGPU_ERR_HIP, (std::string(#expr ": ") + hipGetErrorString(e_)).c_str());               \
        }                                                                                                           \
    } while(0)

template <template <int> class L, typename... Args>
int dispatch_src(int fmt, Args &&... args)
{
    switch(fmt) {
        case SRC_FILL: return L<SRC_FILL>::run(args...);
        case SRC_RGB565: return L<SRC_RGB565>::run(args...);
        case SRC_ARGB8888: return L<SRC_ARGB8888>::run(args...);
        case SRC_RGBA8888: return L<SRC_RGBA8888>::run(args...);
        case SRC_ARGB8565: return L<SRC_ARGB8565>::run(args...);
        default: return L<SRC_COPY>::run(args...);
    }
}

// Calls f(rows-per-phase constant, discard constant) for the run-time pair (nr, discard): the quarter-turn kernels take
// both as template parameters.  The rotation alone (SRC_COPY) stores nothing to dst, so it has the one variant.
template <int SRC, class F>
void with_nr_discard(int nr, bool discard, F && f)
{
    auto go = [&](auto nr_c) {
        if constexpr(SRC != SRC_COPY) {
            if(discard) {
                f(nr_c, std::true_type{});
                return;
            }
        }
        f(nr_c, std::false_type{});
    };
    switch(nr) {
        case 4: go(std::integral_constant<int, 4>{}); break;
        case 2: go(std::integral_constant<int, 2>{}); break;
        default: go(std::integral_constant<int, 8>{}); break;
    }
}

}  // namespace lvgpu
