// lvgpu_kernels.hpp -- HIP kernels (gfx950) for the batched LVGL pixel path.
//
// Data layout in HBM: every frame is a row-major array of 2-byte pixels, frames lie `*_fs` pixels apart.  All fast
// kernels move 16 bytes of destination pixels (8 px) per lane per access; a wavefront (64 lanes) therefore
// touches whole 128-byte lines.  There is no reuse between pixels, so the kernels are pure HBM streams:
//   k_blend_linear : blend (+ optional un-rotated / 180-degree output) over areas that are contiguous in memory
//   k_tile         : blend and/or 90/270-degree rotation; a wave owns a 64x64-pixel tile, each lane an 8x8 block,
//                    transposes it in registers and exchanges it through wave-private LDS so that both the loads
//                    and the rotated stores are 128-byte contiguous per 8 lanes
//   k_tile_col     : the same for heights off the 32-pixel grid: a column of tiles per workgroup, 1-KB output runs
//   k_band         : the same fused with masked fills and maps: bands of whole rows, linear loads, the turn in LDS
//   k_blend_rows   : blend over any rectangle/alignment: 16-byte-aligned destination groups, row edges pixel-exact
//   k_tile_any / k_flush_linear_any : the flush for frames of any size and alignment
#pragma once
#include "lvgpu_px.hpp"

namespace lvgpu {

struct BlendArgs {
    uint16_t * dst;        // first pixel of the (clipped) area in frame 0
    long long dst_fs;      // frame stride, pixels
    int dst_stride;        // row stride, pixels
    const void * src;      // first source pixel of the area in frame 0 (format per template parameter)
    long long src_fs;      // source pixels
    int src_stride;
    const uint8_t * mask;  // first mask byte of the area in frame 0, or nullptr
    long long mask_fs;     // bytes
    int mask_stride;
    int w, h;              // area size
    int n_frames;
    uint32_t color;        // native fill colour
    uint32_t opa;
    int opa_mode;          // OpaMode
    int blend_fn;          // BlendFn
    int swap;              // LV_COLOR_16_SWAP of dst/src
    PremultFill pf;
    const uint32_t * first_nb;  // per frame: scan index of the first non-black pixel (OM_FILL_PREMULT), or nullptr
    uint16_t * out;        // rotated / copied output (frame 0), or nullptr
    long long out_fs;
    int rot;               // LVGPU_ROT_*
    int swap_out;          // byte-swap pixels written to out
    int raw_opa;           // dsc->opa as given (ARGB8565 destination path)
    int raw_mode;          // dsc->blend_mode as given
    int has_mask;          // a per-pixel alpha exists (mask buffer or source alpha)
    int discard_dst;       // fused blend + flush only: the blended pixels go to out alone, dst is left as it is
    int no_aa;             // anti-aliasing off: mask values (and source alpha) are rounded to 0 / 255 as they are loaded
};

// Workgroups are dealt round-robin over the 8 XCDs (blocks b and b+8 share an XCD and its L2).  Remap the block id
// so that the blocks of one XCD walk one contiguous eighth of the batch: neighbouring tiles (which share 128-byte
// lines at their edges) and neighbouring chunks then meet in the same L2.  Bijective for any grid size; affects
// speed only, never results.
__device__ __forceinline__ uint32_t xcd_remap(uint32_t b, uint32_t n_blocks)
{
    const uint32_t q = n_blocks >> 3, r = n_blocks & 7u, x = b & 7u;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}

// ---------------------------------------------------------------------------------------------------------------
// one pixel
// ---------------------------------------------------------------------------------------------------------------
template <class A>  // BlendArgs or a display-list item: anything with opa_mode, opa, blend_fn, swap, pf
__device__ __forceinline__ uint32_t px_core(const A & a, uint32_t d, uint32_t fg, uint32_t m,
                                            uint32_t scan_idx, uint32_t fnb)
{
    uint32_t o;
    switch(a.opa_mode) {
        case OM_CONST: o = a.opa; break;
        case OM_MASK: o = m; break;
        case OM_FILL_MASK_OPA: o = px_opa<OM_FILL_MASK_OPA>(m, a.opa); break;
        case OM_MAP_MASK_OPA: o = px_opa<OM_MAP_MASK_OPA>(m, a.opa); break;
        default: return px_fill_premult(a.pf, d, scan_idx, fnb);
    }
    switch(a.blend_fn) {
        case BF_ADDITIVE: fg = px_blend_fn<BF_ADDITIVE>(fg, d, a.swap); break;
        case BF_SUBTRACTIVE: fg = px_blend_fn<BF_SUBTRACTIVE>(fg, d, a.swap); break;
        case BF_MULTIPLY: fg = px_blend_fn<BF_MULTIPLY>(fg, d, a.swap); break;
        default: break;
    }
    return px_mix(fg, d, o);
}

// The same with the opacity mode fixed at compile time (OM) and the blend function either known to be the normal
// one (BLENDED == false) or chosen at run time.  The mode is uniform over a launch or a display-list item; taking
// the decision out of the per-pixel code removes a handful of scalar branches per pixel.  OM_RUNTIME = decide per pixel.
// OM_OPAQUE = OM_CONST with opa 255 and the normal blend function: lv_color_mix(fg, bg, 255) is fg whatever bg is
// (m5 = 32: ((f - b) * 32 >> 5) + b == f in every field), so the destination is not even read.
constexpr int OM_RUNTIME = -1;
constexpr int OM_OPAQUE = 5;
template <int OM, bool BLENDED, class A>
__device__ __forceinline__ uint32_t px_core_s(const A & a, uint32_t d, uint32_t fg, uint32_t m, uint32_t scan_idx, uint32_t fnb)
{
    if(OM == OM_RUNTIME) return px_core(a, d, fg, m, scan_idx, fnb);
    if(OM == OM_OPAQUE) return fg;
    if(OM == OM_FILL_PREMULT) return px_fill_premult(a.pf, d, scan_idx, fnb);
    if(!BLENDED) return px_mix5(fg, d, px_mix5_of<OM>(m, a.opa));
    const uint32_t o = OM == OM_CONST ? a.opa : px_opa<OM>(m, a.opa);
    if(BLENDED) {
        switch(a.blend_fn) {
            case BF_ADDITIVE: fg = px_blend_fn<BF_ADDITIVE>(fg, d, a.swap); break;
            case BF_SUBTRACTIVE: fg = px_blend_fn<BF_SUBTRACTIVE>(fg, d, a.swap); break;
            case BF_MULTIPLY: fg = px_blend_fn<BF_MULTIPLY>(fg, d, a.swap); break;
            default: break;
        }
    }
    return px_mix(fg, d, o);
}

// Source pixel j (0..7) of an 8-pixel group held in registers: native colour + alpha.
template <int SRC>
__device__ __forceinline__ void src_px(const uint32_t * sw, int j, int swap, uint32_t & fg, uint32_t & alpha)
{
    if(SRC == SRC_RGB565) {
        uint32_t v = (sw[j >> 1] >> ((j & 1) * 16)) & 0xFFFFu;
        fg = swap ? px_swap16(v) : v;
    }
    else if(SRC == SRC_ARGB8888) {
        fg = px_from_argb8888(sw[j]);
        alpha = sw[j] >> 24;
    }
    else if(SRC == SRC_RGBA8888) {
        fg = px_from_rgba8888(sw[j]);
        alpha = sw[j] >> 24;
    }
    else if(SRC == SRC_ARGB8565) {
        const int byte = 3 * j, w0 = byte >> 2, sh = (byte & 3) * 8;
        uint64_t two = (uint64_t)sw[w0] | ((uint64_t)sw[w0 + 1 < 6 ? w0 + 1 : 5] << 32);
        uint32_t v = (uint32_t)(two >> sh);
        uint32_t c = v & 0xFFFFu;
        fg = swap ? px_swap16(c) : c;
        alpha = (v >> 16) & 0xFFu;
    }
}

template <int SRC>
struct SrcWords {
    static constexpr int N = SRC == SRC_RGB565 ? 4 : (SRC == SRC_ARGB8565 ? 6 : (SRC == SRC_FILL || SRC == SRC_COPY ? 1 : 8));
};

// Load the source words of the 8-pixel group that starts at source pixel index `px` (aligned by the launcher).
template <int SRC>
__device__ __forceinline__ void load_src(const void * base, long long px, uint32_t * sw)
{
    if(SRC == SRC_RGB565) {
        uint4 v = *reinterpret_cast<const uint4 *>(reinterpret_cast<const uint16_t *>(base) + px);
        sw[0] = v.x; sw[1] = v.y; sw[2] = v.z; sw[3] = v.w;
    }
    else if(SRC == SRC_ARGB8888 || SRC == SRC_RGBA8888) {
        const uint4 * p = reinterpret_cast<const uint4 *>(reinterpret_cast<const uint32_t *>(base) + px);
        uint4 v0 = p[0], v1 = p[1];
        sw[0] = v0.x; sw[1] = v0.y; sw[2] = v0.z; sw[3] = v0.w;
        sw[4] = v1.x; sw[5] = v1.y; sw[6] = v1.z; sw[7] = v1.w;
    }
    else if(SRC == SRC_ARGB8565) {
        const uint2 * p = reinterpret_cast<const uint2 *>(reinterpret_cast<const uint8_t *>(base) + px * 3);
        uint2 v0 = p[0], v1 = p[1], v2 = p[2];
        sw[0] = v0.x; sw[1] = v0.y; sw[2] = v1.x; sw[3] = v1.y; sw[4] = v2.x; sw[5] = v2.y;
    }
}

// Anti-aliasing off (lv_draw_sw_blend.c:150-156) applied to the words of one 8-pixel group as they sit in registers:
// the mask bytes, and for the formats with their own alpha -- whose alpha plane is the mask of the blend
// (lv_draw_sw_img.c:133-136) -- the alpha byte of every pixel.  The reference rounds the caller's mask buffer in place;
// rounding on the way in gives the same pixels and leaves the buffer alone (rounding is idempotent).
template <int SRC>
__device__ __forceinline__ void no_aa_words(uint32_t * sw, uint2 & mk)
{
    mk.x = px_mask_no_aa_x4(mk.x);
    mk.y = px_mask_no_aa_x4(mk.y);
    if(SRC == SRC_ARGB8888 || SRC == SRC_RGBA8888) {
#pragma unroll
        for(int j = 0; j < 8; j++) sw[j] = (sw[j] & 0x00FFFFFFu) | (px_mask_no_aa_x4(sw[j]) & 0xFF000000u);
    }
    else if(SRC == SRC_ARGB8565) {  // alpha = byte 3j + 2 of the group's 24 bytes
        constexpr uint32_t AM[6] = {0x00FF0000u, 0x0000FF00u, 0xFF0000FFu, 0x00FF0000u, 0x0000FF00u, 0xFF0000FFu};
#pragma unroll
        for(int i = 0; i < 6; i++) sw[i] = (sw[i] & ~AM[i]) | (px_mask_no_aa_x4(sw[i]) & AM[i]);
    }
}

// lv_color_mix over one 8-pixel group for the modes that are nothing but a mix (normal blend function, opacity from
// opa and/or the mask), with the byte order folded into the data movement: one byte permute takes a stored pixel
// out of its dword, exchanges its bytes if the build swaps them and spreads it over both halves
// (lv_color.h:449-450, bg | bg << 16); one more puts two results back side by side in stored order.  Same integer
// arithmetic as px_mix5, pixel by pixel.
template <int SRC, int OM, class A>
__device__ __forceinline__ uint4 blend_group_mix(const A & a, uint4 d, const uint32_t * sw, uint2 mk)
{
    constexpr uint32_t SPREAD = 0x07E0F81Fu;
    const uint32_t dw[4] = {d.x, d.y, d.z, d.w};
    const uint32_t sel_in[2] = {a.swap ? 0x00010001u : 0x01000100u, a.swap ? 0x02030203u : 0x03020302u};
    const uint32_t sel_out = a.swap ? 0x04050001u : 0x05040100u;
    uint32_t rw[4];
#pragma unroll
    for(int j = 0; j < 8; j += 2) {
        uint32_t r[2];
#pragma unroll
        for(int h = 0; h < 2; h++) {
            const int p = j + h;
            const uint32_t b = __builtin_amdgcn_perm(dw[p >> 1], dw[p >> 1], sel_in[h]) & SPREAD;
            uint32_t f, m = ((p < 4 ? mk.x : mk.y) >> ((p & 3) * 8)) & 0xFFu;
            if(SRC == SRC_RGB565) f = __builtin_amdgcn_perm(sw[p >> 1], sw[p >> 1], sel_in[h]) & SPREAD;
            else {
                uint32_t fg = a.color;
                src_px<SRC>(sw, p, a.swap, fg, m);
                f = (fg | (fg << 16)) & SPREAD;
            }
            const uint32_t t = (((f - b) * px_mix5_of<OM>(m, a.opa)) >> 5) + b;
            r[h] = __builtin_amdgcn_bitop3_b32(t >> 16, t, 0x07E0u, 0xE4);  // (t >> 16 & 0x7E0) | (t & ~0x7E0): the pixel in the low half; the high half is not looked at below
        }
        rw[j >> 1] = __builtin_amdgcn_perm(r[1], r[0], sel_out);
    }
    return make_uint4(rw[0], rw[1], rw[2], rw[3]);
}

// The translucent unmasked fill (OM_FILL_PREMULT) over one 8-pixel group, two pixels per packed operation.  Only a
// group that starts before the frame's first non-black pixel can hold pixels that still see the initial cache entry.
template <class A>
__device__ __forceinline__ uint4 blend_group_premult(const A & a, uint4 d, uint32_t scan0, uint32_t fnb)
{
    const uint32_t dw[4] = {d.x, d.y, d.z, d.w};
    uint32_t rw[4];
#pragma unroll
    for(int i = 0; i < 4; i++) {
        const uint32_t n = px_swap16x2_if(dw[i], a.swap);
        uint32_t r = px_mix_premult_x2(a.pf, n);
        if(scan0 < fnb) {
            if((n & 0xFFFFu) == 0u && scan0 + 2u * i < fnb) r = (r & 0xFFFF0000u) | a.pf.first_res;
            if((n >> 16) == 0u && scan0 + 2u * i + 1u < fnb) r = (r & 0x0000FFFFu) | (a.pf.first_res << 16);
        }
        rw[i] = px_swap16x2_if(r, a.swap);
    }
    return make_uint4(rw[0], rw[1], rw[2], rw[3]);
}

// Blend one 8-pixel group.  d: stored destination pixels; mk: 8 mask bytes (ignored for formats with own alpha).
template <int SRC, int OM, bool BLENDED, bool NO_AA_SWITCH = true, class A>
__device__ __forceinline__ uint4 blend_group_s(const A & a, uint4 d, uint32_t * sw, uint2 mk, uint32_t scan0, uint32_t fnb)
{
    if(OM == OM_OPAQUE && SRC == SRC_RGB565) return make_uint4(sw[0], sw[1], sw[2], sw[3]);  // same stored format
    if constexpr(NO_AA_SWITCH && OM != OM_OPAQUE && OM != OM_FILL_PREMULT && OM != OM_CONST) {
        if(a.no_aa) no_aa_words<SRC>(sw, mk);
    }
    if(OM == OM_FILL_PREMULT) return blend_group_premult(a, d, scan0, fnb);
    if(!BLENDED && (OM == OM_CONST || OM == OM_MASK || OM == OM_FILL_MASK_OPA || OM == OM_MAP_MASK_OPA))
        return blend_group_mix<SRC, OM>(a, d, sw, mk);
    uint32_t dw[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
    for(int i = 0; i < 4; i++) dw[i] = px_swap16x2_if(dw[i], a.swap);
    uint32_t rw[4];
#pragma unroll
    for(int j = 0; j < 8; j++) {
        uint32_t dn = (dw[j >> 1] >> ((j & 1) * 16)) & 0xFFFFu;
        uint32_t fg = a.color;
        uint32_t m = ((j < 4 ? mk.x : mk.y) >> ((j & 3) * 8)) & 0xFFu;
        src_px<SRC>(sw, j, a.swap, fg, m);
        uint32_t r = px_core_s<OM, BLENDED>(a, dn, fg, m, scan0 + (uint32_t)j, fnb);
        if(j & 1) rw[j >> 1] |= r << 16;
        else rw[j >> 1] = r;
    }
#pragma unroll
    for(int i = 0; i < 4; i++) rw[i] = px_swap16x2_if(rw[i], a.swap);
    return make_uint4(rw[0], rw[1], rw[2], rw[3]);
}
template <int SRC, class A>
__device__ __forceinline__ uint4 blend_group(const A & a, uint4 d, uint32_t * sw, uint2 mk, uint32_t scan0, uint32_t fnb)
{
    return blend_group_s<SRC, OM_RUNTIME, true, false>(a, d, sw, mk, scan0, fnb);  // the caller has dealt with a.no_aa
}
// The full-frame kernels decide per pixel, except for the two cases whose cost is all memory traffic; the choice is
// uniform over the launch:
//   FAST_MIX     an image with per-pixel alpha (or a masked fill / map) at full opacity under the normal blend function:
//                nothing but lv_color_mix with the pixel's own alpha.  The kernels stay HBM-bound on the whole chip
//                either way; on a fraction of the CUs (32 of 256: +25 %) the shorter instruction stream is what counts.
//   FAST_OPAQUE  fill_normal / map_normal without mask at opa >= LV_OPA_MAX (lv_draw_sw_blend.c:254-259, :655-660:
//                lv_color_fill / lv_memcpy per row): the destination is only written, so it is not loaded either.
//   FAST_PREMULT fill_normal without mask below LV_OPA_MAX (:261-287): the premultiplied formula in packed 16-bit
//                arithmetic; per pixel it is bound by instruction issue (29 -> 15 lane instructions per pixel).
//   FAST_MIX_CONST / _FILL_OPA / _MAP_OPA: the other three ways the normal blend function derives a pixel's opacity
//                (map_normal without mask below LV_OPA_MAX :662-670; masked fill :346-373 and masked map :720-732 below
//                the threshold): still nothing but lv_color_mix, with the opacity mode fixed for the launch.
enum FastPath : int { FAST_NONE = 0, FAST_MIX = 1, FAST_OPAQUE = 2, FAST_PREMULT = 3, FAST_MIX_CONST = 4, FAST_MIX_FILL_OPA = 5,
                      FAST_MIX_MAP_OPA = 6 };
template <int SRC, class A>
__host__ __device__ __forceinline__ int fast_path(const A & a)
{
    constexpr bool own_alpha = SRC == SRC_ARGB8888 || SRC == SRC_RGBA8888 || SRC == SRC_ARGB8565;
    if(SRC == SRC_COPY || a.blend_fn != BF_NORMAL) return FAST_NONE;
    if(a.opa_mode == OM_MASK) return FAST_MIX;
    if(a.opa_mode == OM_FILL_PREMULT) return SRC == SRC_FILL ? FAST_PREMULT : FAST_NONE;
    if(a.opa_mode == OM_CONST && a.opa == 255u && !own_alpha) return FAST_OPAQUE;
    if(a.opa_mode == OM_CONST) return FAST_MIX_CONST;
    if(a.opa_mode == OM_FILL_MASK_OPA) return FAST_MIX_FILL_OPA;
    if(a.opa_mode == OM_MAP_MASK_OPA) return FAST_MIX_MAP_OPA;
    return FAST_NONE;
}
// the launch is nothing but lv_color_mix at an opacity derived one way for all pixels
__host__ __device__ __forceinline__ bool fast_is_mix(int fast)
{
    return fast == FAST_MIX || fast == FAST_MIX_CONST || fast == FAST_MIX_FILL_OPA || fast == FAST_MIX_MAP_OPA;
}
// NO_AA_SWITCH: the kernel honours a.no_aa (the contiguous blend kernel does; the quarter-turn kernels are compiled
// without it -- the branch costs their ARGB8888 instantiations 5 to 34 registers -- and the launcher sends such calls
// through blend + flush as two passes).
template <int SRC, bool NO_AA_SWITCH, class A>
__device__ __forceinline__ uint4 blend_group_u(const A & a, int fast, uint4 d, uint32_t * sw, uint2 mk, uint32_t scan0,
                                               uint32_t fnb)
{
    if constexpr(NO_AA_SWITCH) {
        if(a.no_aa) no_aa_words<SRC>(sw, mk);  // uniform over the launch
    }
    if(fast == FAST_MIX) return blend_group_mix<SRC, OM_MASK>(a, d, sw, mk);
    if(fast == FAST_MIX_CONST) return blend_group_mix<SRC, OM_CONST>(a, d, sw, mk);
    if(fast == FAST_MIX_FILL_OPA) return blend_group_mix<SRC, OM_FILL_MASK_OPA>(a, d, sw, mk);
    if(fast == FAST_MIX_MAP_OPA) return blend_group_mix<SRC, OM_MAP_MASK_OPA>(a, d, sw, mk);
    if(fast == FAST_OPAQUE) return blend_group_s<SRC, OM_OPAQUE, false, false>(a, d, sw, mk, scan0, fnb);
    if(fast == FAST_PREMULT) return blend_group_premult(a, d, scan0, fnb);
    return blend_group<SRC>(a, d, sw, mk, scan0, fnb);
}

// The panel image is written once and never read back by these kernels: a non-temporal store keeps it from
// displacing the draw-buffer lines that are still being read-modified-written in L2 (measured +2 % on the fused path).
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_streaming(uint16_t * p, uint4 v)
{
    u32x4_t t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<u32x4_t *>(p));
}
// The same with the line also dropped from the L2 once written (sc1): for launches that only write their
// destination (opaque fill, opaque map).  A plain device copy moves 81 % of the HBM peak with these stores against
// 78 % with nt alone and 76 % with plain ones; kernels that read what they write gain nothing from it.
__device__ __forceinline__ void store_write_once(uint16_t * p, uint4 v)
{
    u32x4_t t = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(p), "v"(t) : "memory");
}

// The LDS exchange areas are private to one wavefront.  A wave executes its LDS instructions in order, so a read
// that follows a write in program order sees it; all that is needed is to keep the compiler from reordering the
// two, which this does (no s_barrier: the waves of a block never wait for each other).
__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ uint4 swap_group(uint4 v)
{
    return make_uint4(px_swap16x2(v.x), px_swap16x2(v.y), px_swap16x2(v.z), px_swap16x2(v.w));
}
// reverse the order of the 8 pixels of a group
__device__ __forceinline__ uint4 reverse_group(uint4 v)
{
    auto r = [](uint32_t x) { return (x << 16) | (x >> 16); };
    return make_uint4(r(v.w), r(v.z), r(v.y), r(v.x));
}

// ---------------------------------------------------------------------------------------------------------------
// k_blend_linear: the area of every frame is one contiguous run of gpf groups of 8 pixels.
// Block b handles chunk (b % cpf) of frame (b / cpf): LIN_UNROLL groups per lane, loads first, then math, stores.
// out (optional): rot NONE -> same order; rot 180 -> whole-frame reversal (src/lv_port.c:522-527, :533-538).
// ---------------------------------------------------------------------------------------------------------------
constexpr int LIN_UNROLL = 2;
constexpr int LIN_CHUNK = 256 * LIN_UNROLL;  // groups per block

template <int SRC>
__global__ __launch_bounds__(256) void k_blend_linear(BlendArgs a, uint32_t gpf, uint32_t cpf)
{
    const int fast = fast_path<SRC>(a);
    const uint32_t blk = xcd_remap(blockIdx.x, gridDim.x);
    const uint32_t f = blk / cpf;
    const uint32_t c = blk - f * cpf;
    uint16_t * dst = a.dst + (long long)f * a.dst_fs;
    const uint8_t * mask = a.mask ? a.mask + (long long)f * a.mask_fs : nullptr;
    const long long src_px0 = (long long)f * a.src_fs;
    const uint32_t fnb = a.first_nb ? a.first_nb[f] : 0xFFFFFFFFu;
    const uint32_t g0 = c * LIN_CHUNK + threadIdx.x;

    uint4 d[LIN_UNROLL];
    uint2 mk[LIN_UNROLL];
    uint32_t sw[LIN_UNROLL][SrcWords<SRC>::N];
    constexpr bool BLEND = SRC != SRC_COPY;
#pragma unroll
    for(int u = 0; u < LIN_UNROLL; u++) {
        const uint32_t g = g0 + u * 256;
        mk[u] = make_uint2(0, 0);
        if(g < gpf) {
            d[u] = fast == FAST_OPAQUE ? make_uint4(0, 0, 0, 0) : *reinterpret_cast<const uint4 *>(dst + (size_t)g * 8);
            if(BLEND) {
                if(SRC != SRC_FILL) load_src<SRC>(a.src, src_px0 + (long long)g * 8, sw[u]);
                if(mask) mk[u] = *reinterpret_cast<const uint2 *>(mask + (size_t)g * 8);
            }
        }
    }
#pragma unroll
    for(int u = 0; u < LIN_UNROLL; u++) {
        const uint32_t g = g0 + u * 256;
        if(g < gpf) {
            uint4 r = d[u];
            if(BLEND) {
                r = blend_group_u<SRC, true>(a, fast, d[u], sw[u], mk[u], g * 8, fnb);
                if(a.discard_dst) {}
                else if(fast == FAST_OPAQUE) store_write_once(dst + (size_t)g * 8, r);
                else *reinterpret_cast<uint4 *>(dst + (size_t)g * 8) = r;
            }
            if(a.out) {
                uint16_t * out = a.out + (long long)f * a.out_fs;
                if(a.swap_out) r = swap_group(r);
                if(a.rot == 2) store_streaming(out + (size_t)(gpf - 1 - g) * 8, reverse_group(r));
                else store_streaming(out + (size_t)g * 8, r);
            }
        }
    }
}

// The register transpose of the tile kernels.  A lane owns an 8x8-pixel block and holds NR of its rows (phase p of
// 8 / NR: rows p*NR .. p*NR + NR - 1) as 16-byte groups r[0..NR).  Column j of the block becomes an 8-pixel piece of
// a panel row, ordered along that row; NR rows make NR / 2 dwords of it (one v_perm_b32 per dword), which go to the
// 16-byte LDS chunk col0[j * stride] at the place their rows have in the piece: ascending for 270 degrees, descending
// for 90.  With NR = 8 a chunk is written whole; with NR = 4 or 2 a lane never holds more than NR rows of source and
// destination pixels at a time, which is what sets the kernels' register count.
template <int NR>
__device__ __forceinline__ void put_transposed(uint4 * col0, int stride, const uint4 * r, int p, bool r90)
{
    constexpr int NW = NR / 2;
    const int off = r90 ? (8 / NR - 1 - p) * NW : p * NW;
#pragma unroll
    for(int j = 0; j < 8; j++) {
        uint32_t o[NW];
#pragma unroll
        for(int k = 0; k < NW; k++) {
            auto word = [j](const uint4 & v) { return j < 2 ? v.x : (j < 4 ? v.y : (j < 6 ? v.z : v.w)); };
            const uint32_t lo = r90 ? word(r[NR - 1 - 2 * k]) : word(r[2 * k]);
            const uint32_t hi = r90 ? word(r[NR - 2 - 2 * k]) : word(r[2 * k + 1]);
            o[k] = (j & 1) ? __builtin_amdgcn_perm(hi, lo, 0x07060302u) : __builtin_amdgcn_perm(hi, lo, 0x05040100u);
        }
        uint32_t * c = reinterpret_cast<uint32_t *>(col0 + j * stride) + off;
        if(NW == 4) *reinterpret_cast<uint4 *>(c) = make_uint4(o[0], o[NW > 1 ? 1 : 0], o[NW > 2 ? 2 : 0], o[NW > 3 ? 3 : 0]);
        else if(NW == 2) *reinterpret_cast<uint2 *>(c) = make_uint2(o[0], o[NW > 1 ? 1 : 0]);
        else *c = o[0];
    }
}

// ---------------------------------------------------------------------------------------------------------------
// k_tile: whole frames of W x H pixels (W % 8 == 0, H % 8 == 0), optional blend, 90/270-degree output.
//   ROT_90 : out[x*H + (H-1-y)] = px(x,y)      (src/lv_port.c:500-505)
//   ROT_270: out[(W-1-x)*H + y] = px(x,y)      (src/lv_port.c:511-516)
// A wave owns tile (tx,ty) of 64x64 px; lane l owns the 8x8 block (bx = l&7, by = l>>3).  It loads 8 row
// segments of 16 B (8 lanes = 128 B contiguous), blends, stores them back, transposes the block in registers
// (one v_perm per output dword) and writes 8 column segments to the wave's 8 KB of LDS, XOR-swizzled so that
// neither the 16-B writes nor the 16-B reads conflict.  After the exchange lane l holds, for k = 0..7, the segment
// of output row (8k + l>>3) at 16-B chunk (l&7), so 8 consecutive lanes store 128 contiguous bytes.
// ---------------------------------------------------------------------------------------------------------------
constexpr int TILE = 64;
// Wavefronts (= tiles) per workgroup of k_tile.  Two rather than four: the launcher caps how many workgroups share a
// CU, and small workgroups let it do so in steps of two waves (measured: +1 % fused, +1 % rotation alone).
constexpr int TILE_WPB = 2;

// NR: rows of its 8x8 block a lane loads, blends and transposes at a time (8, 4 or 2; see put_transposed).
// DISCARD: the blended pixels are not stored to dst (BlendArgs::discard_dst); a template parameter because a run-time
// test around the stores keeps the compiler from mixing them with the arithmetic (480x320 fused: 69.4 against
// 70.0 % of the HBM peak, 80x160: 53.9 against 58.9 %).
template <int SRC, int NR, bool DISCARD>
__global__ __launch_bounds__(64 * TILE_WPB) void k_tile(BlendArgs a, int W, int H, uint32_t tiles_x, uint32_t tiles_y, uint32_t tpf,
                                              uint32_t n_tiles)
{
    __shared__ uint4 lds[TILE_WPB][512];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const uint32_t wt = __builtin_amdgcn_readfirstlane(xcd_remap(blockIdx.x, gridDim.x) * TILE_WPB + wave);
    const bool live = wt < n_tiles;
    const uint32_t f = live ? wt / tpf : 0;
    const uint32_t t = live ? wt - f * tpf : 0;
    // Tiles of a frame are visited in blocks of 8x8 tiles (512x512 px), row-major inside a block: consecutive waves
    // then read 1 KB runs of the same source rows and complete 1 KB runs of the same panel rows while those lines
    // are still in L2 (matters for canvases much larger than the L2; a 480x320 frame is a single block).
    uint32_t tx, ty;
    {
        const uint32_t sy = t / (8u * tiles_x), r = t - sy * 8u * tiles_x;
        const uint32_t th = min(8u, tiles_y - 8u * sy);
        const uint32_t sx = r / (8u * th), r2 = r - sx * 8u * th;
        const uint32_t tw = min(8u, tiles_x - 8u * sx);
        const uint32_t qy = r2 / tw;
        ty = sy * 8u + qy;
        tx = sx * 8u + (r2 - qy * tw);
    }
    constexpr bool BLEND = SRC != SRC_COPY;

    const int bx = lane & 7, by = lane >> 3;
    const int x0 = tx * TILE + bx * 8, y0 = ty * TILE + by * 8;
    const bool inside = live && x0 < W && y0 < H;

    if(inside) {
        uint16_t * dst = a.dst + (long long)f * a.dst_fs;
        const uint8_t * mask = a.mask ? a.mask + (long long)f * a.mask_fs : nullptr;
        const long long src_px0 = (long long)f * a.src_fs;
        const uint32_t fnb = a.first_nb ? a.first_nb[f] : 0xFFFFFFFFu;
        const int fast = fast_path<SRC>(a);
        const bool r90 = a.rot == 1;
        uint4 * col0 = &lds[wave][bx * 64 + (by ^ bx)];  // column j of the block: chunk (bx*8 + j) * 8 + (by ^ bx)
#pragma unroll
        for(int ph = 0; ph < 8 / NR; ph++) {
            uint4 d[NR], r[NR];
            uint2 mk[NR];
            uint32_t sw[NR][SrcWords<SRC>::N];
#pragma unroll
            for(int i = 0; i < NR; i++) {
                const size_t p = (size_t)(y0 + ph * NR + i) * W + x0;
                d[i] = fast == FAST_OPAQUE ? make_uint4(0, 0, 0, 0) : *reinterpret_cast<const uint4 *>(dst + p);
                mk[i] = make_uint2(0, 0);
                if(BLEND) {
                    if(SRC != SRC_FILL) load_src<SRC>(a.src, src_px0 + (long long)p, sw[i]);
                    if(mask) mk[i] = *reinterpret_cast<const uint2 *>(mask + p);
                }
            }
#pragma unroll
            for(int i = 0; i < NR; i++) {
                r[i] = d[i];
                if(BLEND) {
                    const size_t p = (size_t)(y0 + ph * NR + i) * W + x0;
                    r[i] = blend_group_u<SRC, false>(a, fast, d[i], sw[i], mk[i], (uint32_t)p, fnb);
                    if constexpr(!DISCARD) *reinterpret_cast<uint4 *>(dst + p) = r[i];
                }
                if(a.swap_out) r[i] = swap_group(r[i]);
            }
            put_transposed<NR>(col0, 8, r, ph, r90);
            if(NR < 8) __builtin_amdgcn_sched_barrier(0);  // the next rows are not asked for before these are done with
        }
    }
    __syncthreads();  // a block barrier rather than a wave fence: keeping the block's waves in step measured 2 % faster
    if(live && a.out) {
        uint16_t * out = a.out + (long long)f * a.out_fs;
        const int rr = lane >> 3, c = lane & 7;
        const int yb = ty * TILE + c * 8;  // first source row of the segment
#pragma unroll
        for(int k = 0; k < 8; k++) {
            const int xl = k * 8 + rr;
            const int X = tx * TILE + xl;
            if(X < W && yb < H) {
                uint4 v = lds[wave][xl * 8 + (c ^ k)];
                size_t o = a.rot == 1 ? (size_t)X * H + (size_t)(H - 8 - yb) : (size_t)(W - 1 - X) * H + (size_t)yb;
                store_streaming(out + o, v);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// k_tile_col: k_tile for frames whose height is not a multiple of 32 pixels.  There the panel rows (2*H bytes each)
// start on 32/64/96-byte offsets within a 128-byte line, a single tile's 128-byte output runs fall apart into two
// pieces, and pieces below 64 bytes are written at a fraction of the full-line rate.  Here the wpb wavefronts of a
// workgroup (chosen at launch, 1 to 16; 8 KB of dynamic LDS each) take vertically adjacent tiles (64 columns x 64*wpb
// rows), exchange their transposed blocks through the workgroup's LDS, and every wavefront then writes whole panel-row
// runs of 128*wpb bytes (64 lanes x 16 bytes per store instruction): only the first and the last line of a run are
// partial, and up to 1024 rows one workgroup writes whole panel rows, which follow each other in memory.
// ---------------------------------------------------------------------------------------------------------------
constexpr int COL_WPB_MAX = 16;
// most wavefronts per workgroup for NR rows per phase: with the whole block per phase the blend needs more than the 128
// registers a lane of a 16-wavefront workgroup may have
constexpr int col_wpb_max(int nr) { return nr == 8 ? 8 : COL_WPB_MAX; }
// tpw (1, 2 or 4): tiles per wavefront, one after the other (wavefront w takes tiles w, w + wpb, ...), so that a
// workgroup can be as high as a frame of more than 16 tiles while its LDS (8 KB per tile) allows.
template <int SRC, int NR, bool DISCARD>
__global__ __launch_bounds__(64 * col_wpb_max(NR)) void k_tile_col(BlendArgs a, int W, int H, uint32_t tiles_x, uint32_t cols_y, int wpb,
                                                                 int tpw)
{
    extern __shared__ uint4 lds[];   // [panel row of the block 0..63][chunk of the run 0..RUN-1], swizzled
    const int RUN = 8 * wpb * tpw;   // 16-byte chunks per output run
    const int TH = TILE * wpb * tpw; // rows of the block
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const uint32_t blk = xcd_remap(blockIdx.x, gridDim.x);
    const uint32_t bpf = tiles_x * cols_y;
    const uint32_t f = blk / bpf;
    const uint32_t t = blk - f * bpf;
    // the blocks of a column follow each other: their runs join into whole panel rows while the lines they share are
    // still in L2 (3840x2160: rotation alone 60.7 -> 63.8 %, fused 64.1 -> 64.8 % against row-major block order)
    const uint32_t tx = t / cols_y, cy = t - tx * cols_y;
    constexpr bool BLEND = SRC != SRC_COPY;
    const bool r90 = a.rot == 1;

    const int bx = lane & 7, by = lane >> 3;
    const int Y0 = (int)cy * TH;                               // first source row of the block
    const int x0 = tx * TILE + bx * 8;
    uint16_t * dst = a.dst + (long long)f * a.dst_fs;
    const uint8_t * mask = a.mask ? a.mask + (long long)f * a.mask_fs : nullptr;
    const long long src_px0 = (long long)f * a.src_fs;
    const uint32_t fnb = a.first_nb ? a.first_nb[f] : 0xFFFFFFFFu;
    const int fast = fast_path<SRC>(a);
    for(int s = 0; s < tpw; s++) {
        const int tile = wave + s * wpb;                       // this wavefront's s-th tile of the block
        const int y0 = Y0 + tile * TILE + by * 8;
        if(x0 >= W || y0 >= H) continue;                       // H % 8 == 0: a lane's 8 rows are inside or outside together
        // position of this lane's 8 rows in the run, in ascending panel address order
        const int c = r90 ? RUN - 1 - (tile * 8 + by) : tile * 8 + by;
        uint4 * col0 = &lds[bx * 8 * RUN + (c ^ bx)];          // column j of the block: chunk (bx*8 + j) * RUN + (c ^ bx)
#pragma unroll
        for(int ph = 0; ph < 8 / NR; ph++) {
            uint4 d[NR], r[NR];
            uint2 mk[NR];
            uint32_t sw[NR][SrcWords<SRC>::N];
#pragma unroll
            for(int i = 0; i < NR; i++) {
                const size_t p = (size_t)(y0 + ph * NR + i) * W + x0;
                d[i] = fast == FAST_OPAQUE ? make_uint4(0, 0, 0, 0) : *reinterpret_cast<const uint4 *>(dst + p);
                mk[i] = make_uint2(0, 0);
                if(BLEND) {
                    if(SRC != SRC_FILL) load_src<SRC>(a.src, src_px0 + (long long)p, sw[i]);
                    if(mask) mk[i] = *reinterpret_cast<const uint2 *>(mask + p);
                }
            }
#pragma unroll
            for(int i = 0; i < NR; i++) {
                r[i] = d[i];
                if(BLEND) {
                    const size_t p = (size_t)(y0 + ph * NR + i) * W + x0;
                    r[i] = blend_group_u<SRC, false>(a, fast, d[i], sw[i], mk[i], (uint32_t)p, fnb);
                    if constexpr(!DISCARD) *reinterpret_cast<uint4 *>(dst + p) = r[i];
                }
                if(a.swap_out) r[i] = swap_group(r[i]);
            }
            put_transposed<NR>(col0, RUN, r, ph, r90);
            if(NR < 8) __builtin_amdgcn_sched_barrier(0);
        }
    }
    __syncthreads();
    if(a.out) {
        uint16_t * out = a.out + (long long)f * a.out_fs;
        // first pixel of the run within its panel row; negative for the last block of a column, whose run sticks out
        const int run0 = r90 ? H - Y0 - TH : Y0;
        // pass k of the workgroup stores the chunks (k * wpb + wave) * 64 + lane of the block's 64 x RUN, i.e. 8 / tpw
        // panel rows: chunk qc of panel row k * 8 / tpw + xl0
        const int v = wave * 64 + lane, xl0 = v / RUN;
        const int qc = v - xl0 * RUN;
        const int px = run0 + qc * 8;
        const int rows_per_pass = 8 / tpw;
        if(px >= 0 && px + 8 <= H) {
            for(int k = 0; k < 8 * tpw; k++) {
                const int xl = k * rows_per_pass + xl0;
                const int X = tx * TILE + xl;
                if(X < W) {
                    const int R = r90 ? X : W - 1 - X;
                    store_streaming(out + (size_t)R * H + (size_t)px, lds[xl * RUN + (qc ^ (xl >> 3))]);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// k_band: the 90/270-degree work for frames up to 512 pixels wide whose rows follow each other in memory.  A workgroup
// takes a band of 64 whole rows: one contiguous run of every buffer, which its lanes load group after group like the
// linear kernel does (1 KB per wave instruction, whole 128-byte lines whatever the row pitch -- a 480-pixel row is
// 7.5 lines long, so a tile's row segments start in the middle of a line on every other row), blend, and store back
// the same way.  The quarter turn happens in LDS: the band goes in row by row, 16-byte chunks XOR-swizzled with the
// row's block of 8 so that neither the row-wise 16-byte writes nor the column-wise 2-byte reads meet in a bank, and
// comes out as 8-pixel pieces of panel rows, 8 lanes side by side = 128 bytes of one panel row.
// Measured against k_tile on 4096 frames of 480x320: level for the ARGB8888 image (70.0 against 70.2 % of the HBM
// peak) and for the rotation alone (74.2 against 74.5 %), ahead where an A8 mask plane is read (RGB565 + A8 image
// 69.5 against 66.1 %, masked fill 65.9 against 63.5 %: a tile's mask rows are half lines) -- as long as the blend is
// the plain mix: at opacity 200, which takes the per-pixel opacity arithmetic, its eight-wave workgroups fall to
// 53.6 % where k_tile holds 63.4 %, and an ARGB8565 image runs at 62.9 against 68.2 %.  So it is launched for masked
// fills and maps at full opacity only.  Bands of 32 rows (64-byte pieces of panel rows) reach 65 %, sixteen waves per
// band the same as eight.
// ---------------------------------------------------------------------------------------------------------------
constexpr int BAND_H = 64, BAND_WAVES = 8, BAND_MAX_W = 512;
template <int SRC, bool DISCARD>
__global__ __launch_bounds__(64 * BAND_WAVES) void k_band(BlendArgs a, int W, int H, uint32_t bands)
{
    __shared__ uint4 lds[BAND_H * (BAND_MAX_W / 8)];   // [row of the band][chunk of the row ^ (row >> 3)], 64 chunks per row
    constexpr int NT = 64 * BAND_WAVES, IT = BAND_H * (BAND_MAX_W / 8) / NT;
    constexpr bool BLEND = SRC != SRC_COPY;
    const uint32_t blk = xcd_remap(blockIdx.x, gridDim.x);
    const uint32_t f = blk / bands, b = blk - f * bands;
    const int Y0 = (int)b * BAND_H;
    const int rows = H - Y0 < BAND_H ? H - Y0 : BAND_H;   // H % 8 == 0: whole blocks of 8 rows
    const uint32_t cpr = (uint32_t)W >> 3;                // chunks (8-pixel groups) per row
    const uint32_t n_groups = (uint32_t)rows * cpr;
    const size_t band0 = (size_t)Y0 * W;                  // first pixel of the band within its frame
    uint16_t * dst = a.dst + (long long)f * a.dst_fs + band0;
    const uint8_t * mask = a.mask ? a.mask + (long long)f * a.mask_fs + band0 : nullptr;
    const long long src_px0 = (long long)f * a.src_fs + (long long)band0;
    const uint32_t fnb = a.first_nb ? a.first_nb[f] : 0xFFFFFFFFu;
    const int fast = fast_path<SRC>(a);

    uint4 d[IT];
    uint2 mk[IT];
    uint32_t sw[IT][SrcWords<SRC>::N];
#pragma unroll
    for(int u = 0; u < IT; u++) {
        const uint32_t g = (uint32_t)u * NT + threadIdx.x;
        mk[u] = make_uint2(0, 0);
        d[u] = make_uint4(0, 0, 0, 0);
        if(g < n_groups) {
            if(fast != FAST_OPAQUE) d[u] = *reinterpret_cast<const uint4 *>(dst + (size_t)g * 8);
            if(BLEND) {
                if(SRC != SRC_FILL) load_src<SRC>(a.src, src_px0 + (long long)g * 8, sw[u]);
                if(mask) mk[u] = *reinterpret_cast<const uint2 *>(mask + (size_t)g * 8);
            }
        }
    }
    // row and chunk of group u * NT + threadIdx.x, stepped from one u to the next without dividing again
    uint32_t row = threadIdx.x / cpr, cg = threadIdx.x - row * cpr;
    const uint32_t d_row = NT / cpr, d_cg = NT - d_row * cpr;
#pragma unroll
    for(int u = 0; u < IT; u++) {
        const uint32_t g = (uint32_t)u * NT + threadIdx.x;
        if(g < n_groups) {
            uint4 r = d[u];
            if(BLEND) {
                r = blend_group_u<SRC, false>(a, fast, d[u], sw[u], mk[u], (uint32_t)band0 + g * 8, fnb);
                if constexpr(!DISCARD) *reinterpret_cast<uint4 *>(dst + (size_t)g * 8) = r;
            }
            if(a.swap_out) r = swap_group(r);
            lds[row * (BAND_MAX_W / 8) + (cg ^ (row >> 3))] = r;
        }
        row += d_row;
        cg += d_cg;
        if(cg >= cpr) { cg -= cpr; row++; }
    }
    __syncthreads();
    if(a.out) {
        uint16_t * out = a.out + (long long)f * a.out_fs;
        const uint16_t * l16 = reinterpret_cast<const uint16_t *>(lds);
        const bool r90 = a.rot == 1;
        constexpr uint32_t PPR = BAND_H / 8;              // pieces per panel row
        const uint32_t n_out = (uint32_t)W * PPR;
#pragma unroll
        for(int u = 0; u < IT; u++) {
            const uint32_t q = (uint32_t)u * NT + threadIdx.x;
            const uint32_t X = q / PPR, c = q % PPR;      // panel row (source column), block of 8 source rows
            if(q < n_out && (int)(c * 8u) < rows) {
                const uint32_t base = (c * 8u * (BAND_MAX_W / 8) + ((X >> 3) ^ c)) * 8u + (X & 7u);   // row 8c, column X
                uint32_t w[4];
#pragma unroll
                for(int k = 0; k < 4; k++) {
                    const int r_lo = r90 ? 7 - 2 * k : 2 * k, r_hi = r90 ? 6 - 2 * k : 2 * k + 1;
                    w[k] = (uint32_t)l16[base + r_lo * BAND_MAX_W] | ((uint32_t)l16[base + r_hi * BAND_MAX_W] << 16);
                }
                const size_t o = r90 ? (size_t)X * H + (size_t)(H - 8 - (Y0 + (int)c * 8))
                                     : (size_t)(W - 1 - (int)X) * H + (size_t)(Y0 + (int)c * 8);
                store_streaming(out + o, make_uint4(w[0], w[1], w[2], w[3]));
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// k_blend_rows: any rectangle, any alignment.  A lane owns one 16-byte-aligned group of 8 destination pixels of one
// row of the area: aligned 16-B destination access; source and mask rows start anywhere, so they are fetched with
// under-aligned wide loads, which gfx950 executes as single global_load_dwordx4/x2; of the at most two partial groups
// per row only the area's pixels are stored, so that nothing outside the area is ever written.
// ---------------------------------------------------------------------------------------------------------------
typedef uint32_t u32x4_a2 __attribute__((ext_vector_type(4), aligned(2)));
typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
typedef uint32_t u32x2_a1 __attribute__((ext_vector_type(2), aligned(1)));

template <int SRC>
__device__ __forceinline__ void load_src_unaligned(const void * base, long long px, uint32_t * sw)
{
    if(SRC == SRC_RGB565) {
        u32x4_a2 v = *reinterpret_cast<const u32x4_a2 *>(reinterpret_cast<const uint16_t *>(base) + px);
        sw[0] = v.x; sw[1] = v.y; sw[2] = v.z; sw[3] = v.w;
    }
    else if(SRC == SRC_ARGB8888 || SRC == SRC_RGBA8888) {
        const u32x4_a4 * p = reinterpret_cast<const u32x4_a4 *>(reinterpret_cast<const uint32_t *>(base) + px);
        u32x4_a4 v0 = p[0], v1 = p[1];
        sw[0] = v0.x; sw[1] = v0.y; sw[2] = v0.z; sw[3] = v0.w;
        sw[4] = v1.x; sw[5] = v1.y; sw[6] = v1.z; sw[7] = v1.w;
    }
    else if(SRC == SRC_ARGB8565) {
        const u32x2_a1 * p = reinterpret_cast<const u32x2_a1 *>(reinterpret_cast<const uint8_t *>(base) + px * 3);
        u32x2_a1 v0 = p[0], v1 = p[1], v2 = p[2];
        sw[0] = v0.x; sw[1] = v0.y; sw[2] = v1.x; sw[3] = v1.y; sw[4] = v2.x; sw[5] = v2.y;
    }
}

// The same words gathered pixel by pixel, for the pixels j in [j0, j1) only (the others stay zero): used where the
// wide load would reach outside the rows of the source that belong to the area.
template <int SRC>
__device__ __forceinline__ void load_src_guarded(const void * base, long long px, int j0, int j1, uint32_t * sw)
{
#pragma unroll
    for(int i = 0; i < SrcWords<SRC>::N; i++) sw[i] = 0;
    if(SRC == SRC_RGB565) {
        const uint16_t * p = reinterpret_cast<const uint16_t *>(base) + px;
#pragma unroll
        for(int j = 0; j < 8; j++)
            if(j >= j0 && j < j1) sw[j >> 1] |= (uint32_t)p[j] << ((j & 1) * 16);
    }
    else if(SRC == SRC_ARGB8888 || SRC == SRC_RGBA8888) {
        const uint32_t * p = reinterpret_cast<const uint32_t *>(base) + px;
#pragma unroll
        for(int j = 0; j < 8; j++)
            if(j >= j0 && j < j1) sw[j] = p[j];
    }
    else if(SRC == SRC_ARGB8565) {
        const uint8_t * p = reinterpret_cast<const uint8_t *>(base) + px * 3;
#pragma unroll
        for(int b = 0; b < 24; b++)
            if(b / 3 >= j0 && b / 3 < j1) sw[b >> 2] |= (uint32_t)p[b] << ((b & 3) * 8);
    }
}

// How k_blend_rows maps lanes to groups (filled in by the launcher).
struct RowsMap {
    uint32_t groups_per_row;  // lanes per row of the area: the 16-byte groups a row overlaps (worst case when rows differ)
    long long src_span;       // source pixels from the area's first to its last pixel, + 1: what a wide source load may touch
    long long mask_span;      // the same for the mask, in bytes
};

// Store the pixels of r whose bit is set in pm (bit j = pixel j of the 16-byte group at p), nothing else: 8-, 4- and
// 2-byte pieces, so that a row's first and last group cost at most a handful of stores.
__device__ __forceinline__ void store_group_masked(uint16_t * p, uint4 r, uint32_t pm)
{
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for(int half = 0; half < 2; half++) {
        const uint32_t hm = (pm >> (half * 4)) & 0xFu;
        if(hm == 0xFu) *reinterpret_cast<uint2 *>(p + half * 4) = make_uint2(w[half * 2], w[half * 2 + 1]);
        else {
#pragma unroll
            for(int q = 0; q < 2; q++) {
                const uint32_t dm = (hm >> (q * 2)) & 3u;
                const int px = half * 4 + q * 2;
                if(dm == 3u) *reinterpret_cast<uint32_t *>(p + px) = w[half * 2 + q];
                else if(dm == 1u) p[px] = (uint16_t)w[half * 2 + q];
                else if(dm == 2u) p[px + 1] = (uint16_t)(w[half * 2 + q] >> 16);
            }
        }
    }
}

// OM: the opacity mode fixed at compile time (the launcher picks it; OM_RUNTIME for the blend modes and anything
// else).  Lanes are dealt to groups row by row, so a wavefront holds whole runs of consecutive groups of a few rows,
// first and last group of each row included: the short stores that complete a row's first and last 128-byte line are
// issued by the same wavefront, in the same few instructions, as the 16-byte stores to the rest of those lines, and
// the lines reach memory once.  Every lane takes the same path: aligned 16-byte destination load (for a group that
// sticks out of the area this reads up to 14 bytes beside the area, inside the same aligned 16-byte word), wide source
// and mask loads, the vector blend, and a store of the area's pixels only.  A wide source or mask load that would
// leave the rows that belong to the area (first group of the first row, last group of the last row) is replaced by
// loads of the area's pixels alone.
constexpr int ROWS_UNROLL = 1;  // groups per lane (all loaded before any is blended); 2 measured 2-5 % slower
template <int SRC, int OM>
__global__ __launch_bounds__(256) void k_blend_rows(BlendArgs a, RowsMap map)
{
    const uint32_t n_groups = map.groups_per_row * (uint32_t)a.h;
    const uint32_t idx0 = blockIdx.x * (256 * ROWS_UNROLL) + threadIdx.x;
    if(idx0 >= n_groups) return;
    for(uint32_t f = blockIdx.y; f < (uint32_t)a.n_frames; f += gridDim.y) {
        const uint32_t fnb = a.first_nb ? a.first_nb[f] : 0xFFFFFFFFu;
        uint16_t * grp[ROWS_UNROLL];
        uint32_t pm[ROWS_UNROLL], scan0[ROWS_UNROLL];
        uint4 d[ROWS_UNROLL];
        uint32_t sw[ROWS_UNROLL][SrcWords<SRC>::N];
        uint2 mk[ROWS_UNROLL];
#pragma unroll
        for(int u = 0; u < ROWS_UNROLL; u++) {
            const uint32_t idx = idx0 + (uint32_t)u * 256u;
            pm[u] = 0;
            mk[u] = make_uint2(0, 0);
            if(idx >= n_groups) continue;
            const uint32_t y = idx / map.groups_per_row, k = idx - y * map.groups_per_row;
            uint16_t * row = a.dst + (long long)f * a.dst_fs + (size_t)y * a.dst_stride;
            const int off = (int)((reinterpret_cast<uintptr_t>(row) >> 1) & 7u);  // row start within its 16-byte group
            const int x_lo = (int)k * 8 - off;
            if(x_lo >= a.w) continue;
            const int j0 = x_lo < 0 ? -x_lo : 0, j1 = a.w - x_lo < 8 ? a.w - x_lo : 8;  // pixels [j0, j1) are the area's
            pm[u] = ((1u << j1) - 1u) & ~((1u << j0) - 1u);
            grp[u] = row + x_lo;
            scan0[u] = y * (uint32_t)a.w + (uint32_t)x_lo;
            d[u] = *reinterpret_cast<const uint4 *>(grp[u]);
            if(SRC != SRC_FILL) {
                const long long so = (long long)y * a.src_stride + x_lo;
                if(so >= 0 && so + 8 <= map.src_span) load_src_unaligned<SRC>(a.src, (long long)f * a.src_fs + so, sw[u]);
                else load_src_guarded<SRC>(a.src, (long long)f * a.src_fs + so, j0, j1, sw[u]);
            }
            if(a.mask) {
                const long long mo = (long long)y * a.mask_stride + x_lo;
                const uint8_t * mp = a.mask + (long long)f * a.mask_fs + mo;
                if(mo >= 0 && mo + 8 <= map.mask_span) {
                    u32x2_a1 v = *reinterpret_cast<const u32x2_a1 *>(mp);
                    mk[u] = make_uint2(v.x, v.y);
                }
                else {
#pragma unroll
                    for(int j = 0; j < 8; j++)
                        if(j >= j0 && j < j1) {
                            const uint32_t m = mp[j];
                            if(j < 4) mk[u].x |= m << (j * 8);
                            else mk[u].y |= m << ((j - 4) * 8);
                        }
                }
            }
        }
#pragma unroll
        for(int u = 0; u < ROWS_UNROLL; u++) {
            if(pm[u] == 0) continue;
            const uint4 r = blend_group_s<SRC, OM, OM == OM_RUNTIME>(a, d[u], sw[u], mk[u], scan0[u], fnb);
            if(pm[u] == 0xFFu) *reinterpret_cast<uint4 *>(grp[u]) = r;
            else store_group_masked(grp[u], r, pm[u]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// k_tile_any: the 90/270-degree path (alone or fused with the blend) for frames of ANY width and height, at any
// 2-byte alignment -- 240x135, 172x320, 466x466 panels.  The structure of k_tile_col (the wavefronts of a workgroup take
// vertically adjacent tiles and write whole panel-row runs) with under-aligned 16-byte accesses; the 8-pixel groups
// that stick out of the frame at its right and bottom edge are loaded and stored pixel by pixel, inside the frame.
// ---------------------------------------------------------------------------------------------------------------
typedef uint32_t u32x2_a2 __attribute__((ext_vector_type(2), aligned(2)));
typedef uint32_t u32_a2 __attribute__((aligned(2)));
__device__ __forceinline__ uint4 load16_any(const uint16_t * p)
{
    const u32x4_a2 v = *reinterpret_cast<const u32x4_a2 *>(p);
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void store16_any(uint16_t * p, uint4 v, bool streaming)
{
    u32x4_a2 t = {v.x, v.y, v.z, v.w};
    if(streaming) __builtin_nontemporal_store(t, reinterpret_cast<u32x4_a2 *>(p));
    else *reinterpret_cast<u32x4_a2 *>(p) = t;
}
// store_group_masked for a group that starts at any 2-byte-aligned address
__device__ __forceinline__ void store_group_masked_any(uint16_t * p, uint4 r, uint32_t pm)
{
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for(int half = 0; half < 2; half++) {
        const uint32_t hm = (pm >> (half * 4)) & 0xFu;
        if(hm == 0xFu) {
            u32x2_a2 t = {w[half * 2], w[half * 2 + 1]};
            *reinterpret_cast<u32x2_a2 *>(p + half * 4) = t;
        }
        else {
#pragma unroll
            for(int q = 0; q < 2; q++) {
                const uint32_t dm = (hm >> (q * 2)) & 3u;
                const int px = half * 4 + q * 2;
                if(dm == 3u) *reinterpret_cast<u32_a2 *>(p + px) = w[half * 2 + q];
                else if(dm == 1u) p[px] = (uint16_t)w[half * 2 + q];
                else if(dm == 2u) p[px + 1] = (uint16_t)(w[half * 2 + q] >> 16);
            }
        }
    }
}

// The number of wavefronts per workgroup (wpb, 1..8) is chosen at launch so that a workgroup's 64 * wpb rows fit the
// frame height with as few idle wavefronts as possible; up to 512 rows one workgroup takes a column of tiles whole
// and writes complete panel rows, which follow each other in memory.  The exchange area is 8 KB of dynamic LDS per
// wavefront: [panel row of the region 0..63][chunk of the run 0..8*wpb-1], swizzled.
constexpr int ANY_WPB_MAX = 16;
// most wavefronts per workgroup for NR rows per phase: with the whole block per phase the blend needs more than the 128
// registers a lane of a 16-wavefront workgroup may have
constexpr int any_wpb_max(int nr) { return nr == 8 ? 8 : ANY_WPB_MAX; }
template <int SRC, int NR, bool DISCARD>
__global__ __launch_bounds__(64 * any_wpb_max(NR)) void k_tile_any(BlendArgs a, int W, int H, uint32_t tiles_x, uint32_t cols_y, int wpb)
{
    extern __shared__ uint4 lds[];
    const int RUN = 8 * wpb;               // 16-byte chunks per output run
    const int TH = TILE * wpb;             // rows of the region
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const uint32_t blk = xcd_remap(blockIdx.x, gridDim.x);
    const uint32_t bpf = tiles_x * cols_y;
    const uint32_t f = blk / bpf;
    const uint32_t t = blk - f * bpf;
    const uint32_t tx = t / cols_y, cy = t - tx * cols_y;   // the blocks of a column follow each other (see k_tile_col)
    constexpr bool BLEND = SRC != SRC_COPY;
    const bool r90 = a.rot == 1;

    const int bx = lane & 7, by = lane >> 3;
    const int Y0 = (int)cy * TH;
    const int x0 = tx * TILE + bx * 8, y0 = Y0 + wave * TILE + by * 8;
    if(x0 < W && y0 < H) {
        uint16_t * dst = a.dst + (long long)f * a.dst_fs;
        const uint8_t * mask = a.mask ? a.mask + (long long)f * a.mask_fs : nullptr;
        const long long src_px0 = (long long)f * a.src_fs;
        const uint32_t fnb = a.first_nb ? a.first_nb[f] : 0xFFFFFFFFu;
        const int fast = fast_path<SRC>(a);
        const int nx = W - x0 < 8 ? W - x0 : 8;              // columns of this lane's block that exist
        const bool full_x = nx == 8;
        const uint32_t pmx = (1u << nx) - 1u;
        const int c = r90 ? RUN - 1 - (wave * 8 + by) : wave * 8 + by;
        uint4 * col0 = &lds[bx * 8 * RUN + (c ^ bx)];
#pragma unroll
        for(int ph = 0; ph < 8 / NR; ph++) {
            uint4 d[NR], r[NR];
            uint2 mk[NR];
            uint32_t sw[NR][SrcWords<SRC>::N];
#pragma unroll
            for(int i = 0; i < NR; i++) {
                const int y = y0 + ph * NR + i;
                d[i] = make_uint4(0, 0, 0, 0);
                mk[i] = make_uint2(0, 0);
                if(y >= H) continue;
                const size_t p = (size_t)y * a.dst_stride + x0;
                // A block that sticks out of the frame on the right is still loaded with wide accesses wherever another row
                // follows in the same frame: what they read beyond the row's end is the start of that row (x0 + 8 <= W + 7
                // < 2 W, and rows lie at least W pixels apart in all three buffers); only the frame's last row, and frames
                // narrower than a group, are gathered pixel by pixel.  The surplus pixels are never stored.
                const bool wide = full_x || (W >= 8 && y + 1 < H);
                if(fast != FAST_OPAQUE) {
                    if(wide) d[i] = load16_any(dst + p);
                    else {
                        uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
                        for(int j = 0; j < 8; j++)
                            if(j < nx) w[j >> 1] |= (uint32_t)dst[p + j] << ((j & 1) * 16);
                        d[i] = make_uint4(w[0], w[1], w[2], w[3]);
                    }
                }
                if(BLEND) {
                    if(SRC != SRC_FILL) {
                        const long long sp = src_px0 + (long long)y * a.src_stride + x0;
                        if(wide) load_src_unaligned<SRC>(a.src, sp, sw[i]);
                        else load_src_guarded<SRC>(a.src, sp, 0, nx, sw[i]);
                    }
                    if(mask) {
                        const uint8_t * mp = mask + (size_t)y * a.mask_stride + x0;
                        if(wide) {
                            const u32x2_a1 v = *reinterpret_cast<const u32x2_a1 *>(mp);
                            mk[i] = make_uint2(v.x, v.y);
                        }
                        else {
#pragma unroll
                            for(int j = 0; j < 8; j++)
                                if(j < nx) {
                                    const uint32_t m = mp[j];
                                    if(j < 4) mk[i].x |= m << (j * 8);
                                    else mk[i].y |= m << ((j - 4) * 8);
                                }
                        }
                    }
                }
            }
#pragma unroll
            for(int i = 0; i < NR; i++) {
                const int y = y0 + ph * NR + i;
                r[i] = d[i];
                if(BLEND && y < H) {
                    const size_t p = (size_t)y * a.dst_stride + x0;
                    r[i] = blend_group_u<SRC, false>(a, fast, d[i], sw[i], mk[i], (uint32_t)(y * W + x0), fnb);
                    if constexpr(!DISCARD) {
                        if(full_x) store16_any(dst + p, r[i], false);
                        else store_group_masked_any(dst + p, r[i], pmx);
                    }
                }
                if(a.swap_out) r[i] = swap_group(r[i]);
            }
            put_transposed<NR>(col0, RUN, r, ph, r90);
            if(NR < 8) __builtin_amdgcn_sched_barrier(0);
        }
    }
    __syncthreads();
    if(a.out) {
        uint16_t * out = a.out + (long long)f * a.out_fs;
        const int run0 = r90 ? H - Y0 - TH : Y0;   // first pixel of the run within its panel row (may be < 0)
        // chunk q = (k * wpb + wave) * 64 + lane of the region's 64 x RUN chunks lies in panel row q / RUN = 8 k + xl0
        const int u = wave * 8 + (lane >> 3), xl0 = u / wpb;
        const int qc = (u - xl0 * wpb) * 8 + (lane & 7);
#pragma unroll
        for(int k = 0; k < 8; k++) {
            const int xl = k * 8 + xl0;
            const int X = tx * TILE + xl;
            const int px = run0 + qc * 8;
            if(X >= W || px + 8 <= 0 || px >= H) continue;
            const int k0 = px < 0 ? -px : 0, k1 = H - px < 8 ? H - px : 8;   // pixels [k0, k1) of the chunk exist
            const uint32_t pm = ((1u << k1) - 1u) & ~((1u << k0) - 1u);
            const int R = r90 ? X : W - 1 - X;
            uint16_t * o = out + (long long)R * H + px;
            const uint4 v = lds[xl * RUN + (qc ^ (xl >> 3))];
            if(pm == 0xFFu) store16_any(o, v, true);
            else store_group_masked_any(o, v, pm);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// ARGB8565 destination (3 bytes per pixel: colour lo, colour hi, alpha) -- what lv_draw_sw_blend_basic draws into
// when the display has screen_transp set (lv_draw_sw_blend.c:129-133, 177-184): fill_argb (:448-530) and map_argb
// (:737-874) on top of set_px_argb (:378-402) / set_px_argb_blend (:404-446) and lv_color_mix_with_alpha
// (lv_color.h:530-577).  One pixel per lane, byte accesses.
// ---------------------------------------------------------------------------------------------------------------
// lv_color_mix_with_alpha (lv_color.h:530-577) as one lv_color_mix whose ratio and resulting opacity are selected
// among the reference's four cases: the foreground as it is (mixing at 255 returns it exactly), the background as it
// is (mixing at 0), a plain mix over an opaque background, and the general case with its division.
__device__ __forceinline__ void px_mix_with_alpha(uint32_t bg, uint32_t bg_opa, uint32_t fg, uint32_t fg_opa,
                                                  uint32_t & res, uint32_t & res_opa)
{
    const bool take_fg = fg_opa >= 253u || bg_opa <= 2u;   // :537-541
    const bool take_bg = fg_opa <= 2u;                      // :543-547
    const bool bg_cover = bg_opa >= 253u;                   // :549-552
    const uint32_t opa4 = 255u - (((255u - fg_opa) * (255u - bg_opa)) >> 8);   // :562
    // (fg_opa * 255) / res_opa without the integer-division sequence: the quotient is below 256 and a quotient
    // that is not whole lies at least 1/255 under the next integer, while the reciprocal's error moves the
    // product by less than 1e-4, so adding 0.001 before truncating gives floor() for every operand pair
    // (checked for all 256 x 256 pairs with the reciprocal off by up to two units in the last place)
    const uint32_t ratio4 = (uint32_t)__builtin_fmaf((float)(fg_opa * 255u), __builtin_amdgcn_rcpf((float)opa4), 0.001f) & 0xFFu;
    const uint32_t ratio = take_fg ? 255u : (take_bg ? 0u : (bg_cover ? fg_opa : ratio4));
    res_opa = take_fg ? fg_opa : (take_bg ? bg_opa : (bg_cover ? 255u : opa4));
    res = px_mix(fg, bg, ratio);
}

// One destination pixel of the ARGB8565 path.  a: anything with raw_opa, raw_mode, has_mask, swap.  fill: no source
// image.  fg/m: source colour (native) and mask or source alpha; bg/bg_opa: what the pixel holds.  Returns what to
// store: wo -> the alpha byte becomes `opa`, wc -> the colour becomes `color` (native).
struct ArgbOut {
    uint32_t color, opa;
    bool wc, wo;
};
template <class A>
__device__ __forceinline__ ArgbOut argb_dst_px(const A & a, bool fill, uint32_t fg, uint32_t m, uint32_t bg, uint32_t bg_opa)
{
    ArgbOut r = {0u, 0u, false, false};
    const uint32_t opa = (uint32_t)a.raw_opa;
    const bool blended = !fill && a.raw_mode >= 1 && a.raw_mode <= 3;
    uint32_t o;
    bool plain_store = false;
    if(!a.has_mask) {
        if(fill) {
            o = opa;
            plain_store = opa >= 253u;           // :465-477 writes colour and opa as they are
        }
        else o = opa >= 253u ? 255u : opa;       // :768-819
    }
    else if(fill) {
        if(opa >= 253u) o = m;                   // :494-505
        else {                                   // :507-528
            if(m == 0u) return r;
            o = m == 255u ? opa : (m * opa) >> 8;
        }
    }
    else {
        if(opa > 253u) o = m;                    // :824-843
        else {                                   // :846-871
            if(m == 0u) return r;
            o = m >= 253u ? opa : (opa * m) >> 8;
        }
    }
    if(plain_store) {
        r.color = fg;
        r.opa = opa;
        r.wc = r.wo = true;
    }
    else if(blended) {                           // set_px_argb_blend
        if(bg_opa <= 2u) return r;
        if(o <= 2u) r.color = bg;
        else {
            uint32_t c = a.raw_mode == 1   ? px_blend_fn<BF_ADDITIVE>(fg, bg, a.swap)
                         : a.raw_mode == 2 ? px_blend_fn<BF_SUBTRACTIVE>(fg, bg, a.swap)
                                           : px_blend_fn<BF_MULTIPLY>(fg, bg, a.swap);
            r.color = o == 255u ? c : px_mix(c, bg, o);
        }
        r.wc = true;
    }
    else {                                       // set_px_argb
        px_mix_with_alpha(bg, bg_opa, fg, o, r.color, r.opa);
        r.wo = true;
        r.wc = r.opa > 2u;
    }
    return r;
}

template <int SRC>
__global__ __launch_bounds__(256) void k_blend_argb_dst(BlendArgs a)
{
    const uint32_t area = (uint32_t)a.w * (uint32_t)a.h;
    const uint32_t idx = blockIdx.x * 256 + threadIdx.x;
    if(idx >= area) return;
    const uint32_t y = idx / (uint32_t)a.w, x = idx - y * (uint32_t)a.w;
    const bool fill = SRC == SRC_FILL;
    for(uint32_t f = blockIdx.y; f < (uint32_t)a.n_frames; f += gridDim.y) {
        uint8_t * dp = reinterpret_cast<uint8_t *>(a.dst) + ((long long)f * a.dst_fs + (size_t)y * a.dst_stride + x) * 3;
        uint32_t fg = a.color, m = 0;
        const size_t sp = (size_t)y * a.src_stride + x;
        if(SRC == SRC_RGB565) {
            uint32_t v = (reinterpret_cast<const uint16_t *>(a.src) + (long long)f * a.src_fs)[sp];
            fg = a.swap ? px_swap16(v) : v;
        }
        else if(SRC == SRC_ARGB8888 || SRC == SRC_RGBA8888) {
            uint32_t v = (reinterpret_cast<const uint32_t *>(a.src) + (long long)f * a.src_fs)[sp];
            fg = SRC == SRC_ARGB8888 ? px_from_argb8888(v) : px_from_rgba8888(v);
            m = v >> 24;
        }
        else if(SRC == SRC_ARGB8565) {
            const uint8_t * p = reinterpret_cast<const uint8_t *>(a.src) + ((long long)f * a.src_fs + (long long)sp) * 3;
            uint32_t v = p[0] | (p[1] << 8);
            fg = a.swap ? px_swap16(v) : v;
            m = p[2];
        }
        if(a.mask) {
            // fill_argb's "only the mask matters" loop (:494-505) never adds (mask_stride - w) at the end of a row:
            // row y of the area reads mask[y*w + x]
            const size_t mi = (fill && (uint32_t)a.raw_opa >= 253u) ? (size_t)y * a.w + x : (size_t)y * a.mask_stride + x;
            m = a.mask[(long long)f * a.mask_fs + mi];
        }
        if(a.no_aa) m = px_mask_no_aa(m);
        uint32_t bg = dp[0] | (dp[1] << 8);
        if(a.swap) bg = px_swap16(bg);
        const ArgbOut r = argb_dst_px(a, fill, fg, m, bg, dp[2]);
        if(r.wo) dp[2] = (uint8_t)r.opa;
        if(r.wc) {
            const uint32_t c = a.swap ? px_swap16(r.color) : r.color;
            dp[0] = (uint8_t)(c & 0xFFu);
            dp[1] = (uint8_t)(c >> 8);
        }
    }
}

// How the effective opacity of a pixel of the ARGB8565 path comes about, decided once per launch or list item.
// AM_GENERIC: the blend modes (set_px_argb_blend) and the plain store of an opaque unmasked fill, pixel by pixel
// through argb_dst_px; the others are set_px_argb with o = opa / the mask / the two mask-times-opa forms.
enum ArgbMode : int { AM_GENERIC = 0, AM_CONST = 1, AM_MASK = 2, AM_FILL_MASK_OPA = 3, AM_MAP_MASK_OPA = 4 };
template <class A>
__device__ __forceinline__ int argb_mode(const A & a, bool fill)
{
    const uint32_t opa = (uint32_t)a.raw_opa;
    if(!fill && a.raw_mode >= 1 && a.raw_mode <= 3) return AM_GENERIC;
    if(!a.has_mask) return (fill && opa >= 253u) ? AM_GENERIC : AM_CONST;
    if(fill) return opa >= 253u ? AM_MASK : AM_FILL_MASK_OPA;
    return opa > 253u ? AM_MASK : AM_MAP_MASK_OPA;
}

// Eight ARGB8565 destination pixels held as six dwords (24 bytes): pixel j occupies bytes 3j..3j+2.
template <class T>
__device__ __forceinline__ void argb_unpack8(const uint32_t * dw, T * v)  // v[j]: 24 bits of pixel j (+ garbage above)
{
    v[0] = dw[0];
    v[1] = __builtin_amdgcn_alignbit(dw[1], dw[0], 24);
    v[2] = __builtin_amdgcn_alignbit(dw[2], dw[1], 16);
    v[3] = dw[2] >> 8;
    v[4] = dw[3];
    v[5] = __builtin_amdgcn_alignbit(dw[4], dw[3], 24);
    v[6] = __builtin_amdgcn_alignbit(dw[5], dw[4], 16);
    v[7] = dw[5] >> 8;
}
__device__ __forceinline__ void argb_pack8(const uint32_t * nv, uint32_t * dw)  // nv[j] < 2^24
{
    dw[0] = nv[0] | (nv[1] << 24);
    dw[1] = (nv[1] >> 8) | (nv[2] << 16);
    dw[2] = (nv[2] >> 16) | (nv[3] << 8);
    dw[3] = nv[4] | (nv[5] << 24);
    dw[4] = (nv[5] >> 8) | (nv[6] << 16);
    dw[5] = (nv[6] >> 16) | (nv[7] << 8);
}

// set_px_argb (lv_draw_sw_blend.c:378-402) with lv_color_mix_with_alpha (lv_color.h:530-577) for one stored pixel, in
// the fewest instructions this file knows (the ARGB8565 kernels are bound by instruction issue).
//   v      the pixel as stored: colour in bytes 0-1 (exchanged when sel_* say so), alpha in byte 2, anything in byte 3
//   fgs    the foreground colour, native, spread: (c | c << 16) & 0x07E0F81F
//   o      the foreground's opacity, 0..255
//   sel_dup / sel_out   v_perm selectors: stored colour -> native colour in both halves; {colour, alpha} -> stored
// Returns the new pixel in bits 0-23, bits 24-31 zero.
// The reference's four cases become one lv_color_mix whose 5-bit ratio and resulting alpha are selected:
//   o >= 253 or alpha <= 2      foreground as it is (ratio 32 returns it exactly), alpha o           (:537-541)
//   o <= 2                      background as it is (ratio 0), alpha kept                            (:543-547)
//   alpha >= 253                ratio from o, alpha 255: the general case with the alpha taken as 255 (:549-552)
//   else                        alpha' = 255 - ((255 - o)(255 - alpha) >> 8) = (65535 - (255 - o)(255 - alpha)) >> 8,
//                               ratio = o * 255 / alpha'                                              (:554-575)
// and set_px_argb's "colour only if the new alpha is above 2" (:391) is ratio 0 whenever o <= 2: the new alpha is at
// most 2 only when both o and the old alpha are.
// (o * 255 / alpha' + 4) >> 3, all divisions rounding down, equals floor(o * 31.875 / alpha' + 0.5).  In float: o * 31.875
// is exact, the reciprocal's error moves the product by less than 1e-5, and a value that is not whole lies at least
// 1 / (8 * 255) under the next one, so adding 0.5002 before truncating gives the integer result for every operand pair
// (checked for all 256 x 256 pairs with the reciprocal off by up to two units in the last place, and on the GPU).
__device__ __forceinline__ uint32_t argb_px_normal(uint32_t v, uint32_t fgs, uint32_t o, uint32_t sel_dup, uint32_t sel_out,
                                                   bool skip = false)
{
    const uint32_t bgs = __builtin_amdgcn_perm(0u, v, sel_dup) & 0x07E0F81Fu;
    const uint32_t b = __builtin_amdgcn_ubfe(v, 16, 8);
    // 65535 - (255 - o)(255 - alpha) = (510 + 255 o) + (255 - o) alpha, alpha taken as 255 when it covers: two v_mad_u32_u24
    const uint32_t t = __umul24(o ^ 255u, b > 252u ? 255u : b) + (__umul24(o, 255u) + 510u);   // 510 <= t <= 65535
    const uint32_t opa4 = t >> 8;
    uint32_t m5d = (uint32_t)__builtin_fmaf((float)o * 31.875f, __builtin_amdgcn_rcpf((float)((t >> 8) & 0xFFu)), 0.5002f);
    // computed for every pixel: as an operand of the selects below the compiler would put the division under a branch
    // per pixel, which costs more than it saves and keeps the pixels of a group from being worked on side by side
    asm volatile("" : "+v"(m5d));
    // skip: the caller wants the pixel left as it is and passes o = 0: ratio 0 keeps the colour, and with the
    // foreground case switched off the alpha is kept too
    const bool o_lo = o < 3u;
    const bool take_fg = (o > 252u) | ((b < 3u) & !skip);
    const uint32_t m5 = o_lo ? 0u : (take_fg ? 32u : m5d);
    const uint32_t res_opa = take_fg ? o : (o_lo ? b : opa4);
    const uint32_t r = (((fgs - bgs) * m5) >> 5) + bgs;
    const uint32_t folded = __builtin_amdgcn_bitop3_b32(r >> 16, r, 0x07E0u, 0xE4);      // (r >> 16 & 0x7E0) | (r & ~0x7E0); bits 16-31 are dropped below
    return __builtin_amdgcn_perm(res_opa, folded, sel_out);
}

// The same over the group for the normal blend mode, opacity per AM: no branches, every pixel independent of its
// neighbours.
template <int SRC, int AM, class A>
__device__ __forceinline__ void argb_group_normal(const A & a, uint32_t * dw, const uint32_t * sw, uint2 mk, uint32_t color)
{
    const uint32_t opa = (uint32_t)a.raw_opa;
    const uint32_t sel_dup = a.swap ? 0x00010001u : 0x01000100u;  // native colour in both halves of the word
    const uint32_t sel_out = a.swap ? 0x0C040001u : 0x0C040100u;  // colour bytes as stored, alpha, zero
    const uint32_t color_s = (color | (color << 16)) & 0x07E0F81Fu;
    uint32_t v[8], nv[8];
    argb_unpack8(dw, v);
#pragma unroll
    for(int j = 0; j < 8; j++) {
        uint32_t fg = color;
        uint32_t m = ((j < 4 ? mk.x : mk.y) >> ((j & 3) * 8)) & 0xFFu;
        src_px<SRC>(sw, j, a.swap, fg, m);
        const uint32_t fgs = SRC == SRC_FILL ? color_s : (fg | (fg << 16)) & 0x07E0F81Fu;
        uint32_t o;
        bool skip = false;
        if(AM == AM_CONST) o = SRC == SRC_FILL ? opa : (opa >= 253u ? 255u : opa);       // :480-488 / :768-819
        else if(AM == AM_MASK) o = m;                                                     // :494-505 / :824-843
        else if(AM == AM_FILL_MASK_OPA) {                                                 // :507-528
            skip = m == 0u;                                                               // then o = 0 as well
            o = m == 255u ? opa : (m * opa) >> 8;
        }
        else {                                                                            // :846-871
            skip = m == 0u;
            o = m >= 253u ? opa : (opa * m) >> 8;
        }
        nv[j] = argb_px_normal(v[j], fgs, o, sel_dup, sel_out, skip);
    }
    argb_pack8(nv, dw);
}

template <int SRC, bool NO_AA_SWITCH = false, class A>
__device__ __forceinline__ void argb_group(const A & a, uint32_t * dw, uint32_t * sw, uint2 mk, uint32_t color)
{
    constexpr bool fill = SRC == SRC_FILL;
    if constexpr(NO_AA_SWITCH) {
        if(a.no_aa) no_aa_words<SRC>(sw, mk);
    }
    switch(argb_mode(a, fill)) {  // the same for every pixel of a launch / list item
        case AM_CONST: argb_group_normal<SRC, AM_CONST>(a, dw, sw, mk, color); return;
        case AM_MASK: argb_group_normal<SRC, AM_MASK>(a, dw, sw, mk, color); return;
        case AM_FILL_MASK_OPA: argb_group_normal<SRC, AM_FILL_MASK_OPA>(a, dw, sw, mk, color); return;
        case AM_MAP_MASK_OPA: argb_group_normal<SRC, AM_MAP_MASK_OPA>(a, dw, sw, mk, color); return;
        default: break;
    }
    uint32_t v[8], nv[8];
    argb_unpack8(dw, v);
#pragma unroll
    for(int j = 0; j < 8; j++) {
        uint32_t bg = v[j] & 0xFFFFu;
        if(a.swap) bg = px_swap16(bg);
        const uint32_t bg_opa = (v[j] >> 16) & 0xFFu;
        uint32_t fg = color;
        uint32_t m = ((j < 4 ? mk.x : mk.y) >> ((j & 3) * 8)) & 0xFFu;
        src_px<SRC>(sw, j, a.swap, fg, m);
        const ArgbOut r = argb_dst_px(a, fill, fg, m, bg, bg_opa);
        const uint32_t c = a.swap ? px_swap16(r.color) : r.color;
        nv[j] = (r.wc ? c : (v[j] & 0xFFFFu)) | ((r.wo ? r.opa : bg_opa) << 16);
    }
    argb_pack8(nv, dw);
}

// ARGB8565 destination over areas that are contiguous in memory.  A lane blends 16 pixels = 48 bytes of destination,
// a wavefront 1024 pixels = 3072 bytes.  48-byte pieces do not make coalesced accesses (each of a lane's three
// 16-byte words would sit 48 bytes from its neighbour's), so the wavefront moves its 3072 bytes as three fully
// coalesced 1-KB instructions -- lane l takes 16-byte word i*64 + l -- and the words change owners through 3 KB of
// wave-private LDS: word c of the run belongs to lane c / 3 (reads at a 48-byte lane stride are conflict-free: 3 is
// odd).  The way back is the same in reverse.  gpf = 16-pixel groups per frame, cpf = blocks per frame.
constexpr int ARGB_WAVE_GROUPS = 64;                    // 16-pixel groups per wavefront
constexpr int ARGB_BLOCK_GROUPS = 4 * ARGB_WAVE_GROUPS; // per 256-thread block
template <int SRC>
__global__ __launch_bounds__(256) void k_blend_argb_linear(BlendArgs a, uint32_t gpf, uint32_t cpf)
{
    __shared__ uint4 xch[4][3 * 64];
    const uint32_t blk = xcd_remap(blockIdx.x, gridDim.x);
    const uint32_t f = blk / cpf;
    const uint32_t c = blk - f * cpf;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint8_t * dst = reinterpret_cast<uint8_t *>(a.dst) + (long long)f * a.dst_fs * 3;
    const uint8_t * mask = a.mask ? a.mask + (long long)f * a.mask_fs : nullptr;
    const long long src_px0 = (long long)f * a.src_fs;
    const uint32_t gw = c * ARGB_BLOCK_GROUPS + (uint32_t)wave * ARGB_WAVE_GROUPS;  // the wavefront's first group
    if(gw >= gpf) return;   // no block-wide barrier below: a wave may leave on its own
    const uint32_t g = gw + (uint32_t)lane;                                          // this lane's group
    const uint32_t n_words = gpf * 3u;                                               // 16-byte words per frame
    uint4 * dst16 = reinterpret_cast<uint4 *>(dst);
    // without a per-pixel alpha and at full opacity colour and alpha are stored whatever the pixel held (:465-477, :768-790)
    const bool blended = SRC != SRC_FILL && a.raw_mode >= 1 && a.raw_mode <= 3;
    const bool need_dst = a.has_mask || a.raw_opa < 253 || blended;
    if constexpr(SRC == SRC_FILL) {
        if(!need_dst) {
            // Opaque unmasked fill (:465-477): every pixel becomes {colour, opa}, a pattern of three bytes.  A 16-byte word
            // of the frame starts 16 * word = word (mod 3) bytes into it and is one of three constants; no exchange.
            const uint32_t c = a.swap ? px_swap16(a.color) : a.color;
            const uint32_t p = c | ((uint32_t)a.raw_opa << 16);
            const uint32_t d0 = p | (p << 24), d1 = (p >> 8) | (p << 16), d2 = (p >> 16) | (p << 8);
            const int l3 = lane % 3;
#pragma unroll
            for(int i = 0; i < 3; i++) {
                const uint32_t word = gw * 3u + (uint32_t)(i * 64 + lane);
                const int ph = (l3 + i) % 3;
                const uint4 w = ph == 0 ? make_uint4(d0, d1, d2, d0) : (ph == 1 ? make_uint4(d1, d2, d0, d1) : make_uint4(d2, d0, d1, d2));
                if(word < n_words) store_write_once(reinterpret_cast<uint16_t *>(dst16 + word), w);
            }
            return;
        }
    }
    uint32_t dw[2][6];
    uint2 mk[2] = {make_uint2(0, 0), make_uint2(0, 0)};
    uint32_t sw[2][SrcWords<SRC>::N];
    if(need_dst) {
        uint4 v[3];
#pragma unroll
        for(int i = 0; i < 3; i++) {
            const uint32_t word = gw * 3u + (uint32_t)(i * 64 + lane);
            v[i] = word < n_words ? dst16[word] : make_uint4(0, 0, 0, 0);
        }
        if(g < gpf) {
            if(SRC != SRC_FILL) {
                load_src<SRC>(a.src, src_px0 + (long long)g * 16, sw[0]);
                load_src<SRC>(a.src, src_px0 + (long long)g * 16 + 8, sw[1]);
            }
            if(mask) {
                const uint4 m = *reinterpret_cast<const uint4 *>(mask + (size_t)g * 16);
                mk[0] = make_uint2(m.x, m.y);
                mk[1] = make_uint2(m.z, m.w);
            }
        }
#pragma unroll
        for(int i = 0; i < 3; i++) xch[wave][i * 64 + lane] = v[i];
        wave_lds_fence();
        const uint4 w0 = xch[wave][3 * lane], w1 = xch[wave][3 * lane + 1], w2 = xch[wave][3 * lane + 2];
        dw[0][0] = w0.x; dw[0][1] = w0.y; dw[0][2] = w0.z; dw[0][3] = w0.w; dw[0][4] = w1.x; dw[0][5] = w1.y;
        dw[1][0] = w1.z; dw[1][1] = w1.w; dw[1][2] = w2.x; dw[1][3] = w2.y; dw[1][4] = w2.z; dw[1][5] = w2.w;
        wave_lds_fence();
    }
    else {
#pragma unroll
        for(int i = 0; i < 6; i++) dw[0][i] = dw[1][i] = 0;
        if(g < gpf && SRC != SRC_FILL) {
            load_src<SRC>(a.src, src_px0 + (long long)g * 16, sw[0]);
            load_src<SRC>(a.src, src_px0 + (long long)g * 16 + 8, sw[1]);
        }
    }
    if(g < gpf) {
        argb_group<SRC, true>(a, dw[0], sw[0], mk[0], a.color);
        argb_group<SRC, true>(a, dw[1], sw[1], mk[1], a.color);
    }
    xch[wave][3 * lane] = make_uint4(dw[0][0], dw[0][1], dw[0][2], dw[0][3]);
    xch[wave][3 * lane + 1] = make_uint4(dw[0][4], dw[0][5], dw[1][0], dw[1][1]);
    xch[wave][3 * lane + 2] = make_uint4(dw[1][2], dw[1][3], dw[1][4], dw[1][5]);
    wave_lds_fence();
#pragma unroll
    for(int i = 0; i < 3; i++) {
        const uint32_t word = gw * 3u + (uint32_t)(i * 64 + lane);
        if(word < n_words) {
            if(need_dst) dst16[word] = xch[wave][i * 64 + lane];
            else store_write_once(reinterpret_cast<uint16_t *>(dst16 + word), xch[wave][i * 64 + lane]);  // a launch that only writes
        }
    }
}

// ARGB8565 destination over any rectangle, eight pixels per lane: k_blend_rows' dealing of lanes to groups row by row,
// for pixels of three bytes.  Nothing about such a row is aligned (its first byte is at 3 * x), so every access is an
// under-aligned wide one: 16 + 8 bytes of destination, 8 of mask, the source as in k_blend_rows.  A row's groups are
// counted from the area's first pixel; when the width is not a multiple of 8 the last group of a row is moved back so
// that it ends with the row, and the pixels it shares with the group before it are blended and stored by both lanes,
// to the same values: the two lanes are neighbours in one wavefront, whose loads all precede its stores.  That they
// are neighbours is arranged by the numbering: a row takes an even number of lanes (map.groups_per_row, one idle lane
// at its start when the number of groups is odd), so a row's last lane has an odd number and never starts a wavefront.
// Areas narrower than 8 pixels are gathered and stored byte by byte.
typedef uint32_t u32x4_a1 __attribute__((ext_vector_type(4), aligned(1)));
template <int SRC>
__global__ __launch_bounds__(256) void k_blend_argb_rows(BlendArgs a, RowsMap map)
{
    const uint32_t n_lanes = map.groups_per_row * (uint32_t)a.h;
    const uint32_t idx = blockIdx.x * 256 + threadIdx.x;
    if(idx >= n_lanes) return;
    constexpr bool fill = SRC == SRC_FILL;
    const uint32_t y = idx / map.groups_per_row;
    const int k = (int)(idx - y * map.groups_per_row) - (int)(map.groups_per_row - (uint32_t)(a.w + 7) / 8u);
    if(k < 0) return;                                  // the idle lane of a row with an odd number of groups
    const bool wide = a.w >= 8;
    const int x_lo = wide ? min(k * 8, a.w - 8) : 0;
    const bool blended = !fill && a.raw_mode >= 1 && a.raw_mode <= 3;
    const bool need_dst = a.has_mask || a.raw_opa < 253 || blended;   // else colour and alpha are stored whatever the pixel held
    // fill_argb's "only the mask matters" loop (:494-505) never adds (mask_stride - w) at the end of a row
    const long long mo = (fill && a.raw_opa >= 253 ? (long long)y * a.w : (long long)y * a.mask_stride) + x_lo;
    const long long so = (long long)y * a.src_stride + x_lo;
    for(uint32_t f = blockIdx.y; f < (uint32_t)a.n_frames; f += gridDim.y) {
        uint8_t * dp = reinterpret_cast<uint8_t *>(a.dst) + ((long long)f * a.dst_fs + (long long)y * a.dst_stride + x_lo) * 3;
        uint32_t dw[6] = {0, 0, 0, 0, 0, 0};
        uint32_t sw[SrcWords<SRC>::N];
        uint2 mk = make_uint2(0, 0);
        if(wide) {
            if(need_dst) {
                const u32x4_a1 v0 = *reinterpret_cast<const u32x4_a1 *>(dp);
                const u32x2_a1 v1 = *reinterpret_cast<const u32x2_a1 *>(dp + 16);
                dw[0] = v0.x; dw[1] = v0.y; dw[2] = v0.z; dw[3] = v0.w; dw[4] = v1.x; dw[5] = v1.y;
            }
            if(!fill) load_src_unaligned<SRC>(a.src, (long long)f * a.src_fs + so, sw);
            if(a.mask) {
                const u32x2_a1 v = *reinterpret_cast<const u32x2_a1 *>(a.mask + (long long)f * a.mask_fs + mo);
                mk = make_uint2(v.x, v.y);
            }
            argb_group<SRC, true>(a, dw, sw, mk, a.color);
            const u32x4_a1 v0 = {dw[0], dw[1], dw[2], dw[3]};
            const u32x2_a1 v1 = {dw[4], dw[5]};
            *reinterpret_cast<u32x4_a1 *>(dp) = v0;
            *reinterpret_cast<u32x2_a1 *>(dp + 16) = v1;
        }
        else {
            const int nb = 3 * a.w;                    // bytes of the row, at most 21
            if(need_dst) {
#pragma unroll
                for(int b = 0; b < 21; b++)
                    if(b < nb) dw[b >> 2] |= (uint32_t)dp[b] << ((b & 3) * 8);
            }
            if(!fill) load_src_guarded<SRC>(a.src, (long long)f * a.src_fs + so, 0, a.w, sw);
            if(a.mask) {
                const uint8_t * mp = a.mask + (long long)f * a.mask_fs + mo;
#pragma unroll
                for(int j = 0; j < 7; j++)
                    if(j < a.w) {
                        const uint32_t m = mp[j];
                        if(j < 4) mk.x |= m << (j * 8);
                        else mk.y |= m << ((j - 4) * 8);
                    }
            }
            argb_group<SRC, true>(a, dw, sw, mk, a.color);
#pragma unroll
            for(int b = 0; b < 21; b++)
                if(b < nb) dp[b] = (uint8_t)(dw[b >> 2] >> ((b & 3) * 8));
        }
    }
}

// Scan index of the first non-black destination pixel of the area, per frame.  Block (g, f) walks the g-th segment
// of seg_px pixels of frame f in scan order, 1024 pixels at a time, and stops at the first piece that holds a
// non-black pixel: for almost every frame that is the first piece of segment 0, so the pass reads a few KB per frame.
// One atomicMin per block that found something (first_nb[] starts at 0xFFFFFFFF = "none").
// (The kernels that are not templates are static: this header is included by several translation units.)
static __global__ __launch_bounds__(256) void k_first_nonblack(const uint16_t * dst, long long dst_fs, int dst_stride, int w, int h,
                                                        int n_frames, uint32_t seg_px, uint32_t * first_nb)
{
    __shared__ uint32_t wave_best[4];
    const uint32_t area = (uint32_t)w * (uint32_t)h;
    const uint32_t lo = blockIdx.x * seg_px, hi = min(lo + seg_px, area);
    const bool contiguous = dst_stride == w;
    for(uint32_t f = blockIdx.y; f < (uint32_t)n_frames; f += gridDim.y) {
        const uint16_t * frame = dst + (long long)f * dst_fs;
        uint32_t found = 0xFFFFFFFFu;
        for(uint32_t base = lo; base < hi; base += 1024u) {  // every thread of the block takes the same trips
            uint32_t best = 0xFFFFFFFFu;
#pragma unroll
            for(int k = 3; k >= 0; k--) {  // descending, so that the smallest index is assigned last
                const uint32_t idx = base + (uint32_t)k * 256u + threadIdx.x;
                if(idx < hi) {
                    size_t at = idx;
                    if(!contiguous) {
                        const uint32_t y = idx / (uint32_t)w;
                        at = (size_t)y * dst_stride + (idx - y * (uint32_t)w);
                    }
                    if(frame[at] != 0) best = idx;
                }
            }
#pragma unroll
            for(int s = 32; s > 0; s >>= 1) best = min(best, (uint32_t)__shfl_xor((int)best, s, 64));
            if((threadIdx.x & 63) == 0) wave_best[threadIdx.x >> 6] = best;
            __syncthreads();
            const uint32_t b = min(min(wave_best[0], wave_best[1]), min(wave_best[2], wave_best[3]));
            __syncthreads();
            if(b != 0xFFFFFFFFu) {
                found = b;
                break;
            }
        }
        if(threadIdx.x == 0 && found != 0xFFFFFFFFu) atomicMin(&first_nb[f], found);
    }
}

// Copy (rot 0) or half turn (rot 2) into the panel image for frames whose pixel count is not a multiple of 8 or whose
// buffers are not 16-byte aligned (466x466 panels; frames packed at odd strides).  A frame's pixels are contiguous, so
// group g of the output is pixels 8g..8g+7 of the input or their mirror image: 8 pixels per lane with under-aligned
// 16-byte accesses; the last, short group of a frame moves pixel by pixel.
static __global__ __launch_bounds__(256) void k_flush_linear_any(const uint16_t * in, long long in_fs, uint16_t * out, long long out_fs,
                                                          uint32_t area, int rot, int swap_out, int n_frames)
{
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;
    const uint32_t o0 = g * 8u;
    if(o0 >= area) return;
    const uint32_t n = min(8u, area - o0);
    for(uint32_t f = blockIdx.y; f < (uint32_t)n_frames; f += gridDim.y) {
        const uint16_t * src = in + (long long)f * in_fs;
        uint16_t * dst = out + (long long)f * out_fs + o0;
        if(n == 8u) {
            uint4 v = rot == 2 ? reverse_group(load16_any(src + (area - 8u - o0))) : load16_any(src + o0);
            if(swap_out) v = swap_group(v);
            store16_any(dst, v, true);
        }
        else {
            for(uint32_t j = 0; j < n; j++) {
                uint32_t v = rot == 2 ? src[area - 1u - o0 - j] : src[o0 + j];
                if(swap_out) v = px_swap16(v);
                dst[j] = (uint16_t)v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// colour conversion, split, copy, synthetic data, checksum
// ---------------------------------------------------------------------------------------------------------------
static __global__ void k_from_rgb888(const uint32_t * in, uint16_t * out, size_t n, int swap)
{
    for(size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        uint32_t c = px_from_argb8888(in[i]);
        out[i] = (uint16_t)(swap ? px_swap16(c) : c);
    }
}
static __global__ void k_to32(const uint16_t * in, uint32_t * out, size_t n, int swap)
{
    for(size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        uint32_t c = in[i];
        out[i] = px_to32(swap ? px_swap16(c) : c);
    }
}
// convert_cb + recolor: fmt SRC_RGB565 copies the colour (alpha 255, or 0 where it equals the chroma key); rc.opa_inv
// == 255 means "no recolor" (lv_color_mix_premult with nothing premultiplied and mix 255 would not be the identity
// in general, so it is skipped as the reference skips it).
static __global__ void k_image_convert(const void * src, int fmt, uint16_t * color, uint8_t * alpha, size_t n, int swap, int keyed,
                                uint32_t key, int recolor, PremultFill rc)
{
    for(size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        uint32_t c, al;  // c: stored lv_color_t
        if(fmt == SRC_RGB565) {
            c = reinterpret_cast<const uint16_t *>(src)[i];
            al = (keyed && c == key) ? 0u : 255u;
        }
        else if(fmt == SRC_ARGB8565) {
            const uint8_t * p = reinterpret_cast<const uint8_t *>(src) + i * 3;
            c = p[0] | (p[1] << 8);
            al = p[2];
        }
        else {
            uint32_t v = reinterpret_cast<const uint32_t *>(src)[i];
            c = fmt == SRC_ARGB8888 ? px_from_argb8888(v) : px_from_rgba8888(v);
            if(swap) c = px_swap16(c);
            al = v >> 24;
        }
        if(recolor) {
            uint32_t nat = swap ? px_swap16(c) : c;
            nat = px_mix_premult(rc, nat);
            c = swap ? px_swap16(nat) : nat;
        }
        color[i] = (uint16_t)c;
        alpha[i] = (uint8_t)al;
    }
}
// Vector forms of the conversions above: a lane converts 8 consecutive pixels with 16-byte accesses (alpha: 8 bytes);
// n8 = number of 8-pixel groups, all pointers aligned to their access size.  The blocks of an XCD take one contiguous
// eighth of the buffer.  What is left over (fewer than 8 pixels, or buffers that are not aligned) takes the scalar kernels.
static __global__ __launch_bounds__(256) void k_from_rgb888_v(const uint4 * in, uint4 * out, size_t n8, int swap)
{
    for(size_t g = xcd_remap(blockIdx.x, gridDim.x) * (size_t)256 + threadIdx.x; g < n8; g += (size_t)gridDim.x * 256) {
        const uint4 a = in[2 * g], b = in[2 * g + 1];
        const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        uint32_t o[4];
#pragma unroll
        for(int i = 0; i < 4; i++) o[i] = px_swap16x2_if(px_from_argb8888(w[2 * i]) | (px_from_argb8888(w[2 * i + 1]) << 16), swap);
        out[g] = make_uint4(o[0], o[1], o[2], o[3]);
    }
}
static __global__ __launch_bounds__(256) void k_to32_v(const uint4 * in, uint4 * out, size_t n8, int swap)
{
    for(size_t g = xcd_remap(blockIdx.x, gridDim.x) * (size_t)256 + threadIdx.x; g < n8; g += (size_t)gridDim.x * 256) {
        const uint4 v = in[g];
        const uint32_t w[4] = {px_swap16x2_if(v.x, swap), px_swap16x2_if(v.y, swap), px_swap16x2_if(v.z, swap), px_swap16x2_if(v.w, swap)};
        uint32_t o[8];
#pragma unroll
        for(int i = 0; i < 4; i++) {
            o[2 * i] = px_to32(w[i] & 0xFFFFu);
            o[2 * i + 1] = px_to32(w[i] >> 16);
        }
        out[2 * g] = make_uint4(o[0], o[1], o[2], o[3]);
        out[2 * g + 1] = make_uint4(o[4], o[5], o[6], o[7]);
    }
}
// convert_cb + recolor (and, without key and recolor, the plain colour/alpha split); key_native: the chroma key as a
// native value (the comparison is between stored pixels, and byte order is a bijection)
template <int SRC>
__global__ __launch_bounds__(256) void k_image_convert_v(const void * src, uint4 * color, uint2 * alpha, size_t n8, int swap, int keyed,
                                                         uint32_t key_native, int recolor, PremultFill rc)
{
    for(size_t g = xcd_remap(blockIdx.x, gridDim.x) * (size_t)256 + threadIdx.x; g < n8; g += (size_t)gridDim.x * 256) {
        uint32_t sw[SrcWords<SRC>::N];
        load_src<SRC>(src, (long long)g * 8, sw);
        uint32_t c[4], al[2] = {0, 0};
#pragma unroll
        for(int j = 0; j < 8; j++) {
            uint32_t fg = 0, m = 255u;
            src_px<SRC>(sw, j, swap, fg, m);
            if(SRC == SRC_RGB565) m = (keyed && fg == key_native) ? 0u : 255u;
            if(j & 1) c[j >> 1] |= fg << 16;
            else c[j >> 1] = fg;
            al[j >> 2] |= m << ((j & 3) * 8);
        }
#pragma unroll
        for(int i = 0; i < 4; i++) {
            if(recolor) c[i] = px_mix_premult_x2(rc, c[i]);
            c[i] = px_swap16x2_if(c[i], swap);
        }
        color[g] = make_uint4(c[0], c[1], c[2], c[3]);
        alpha[g] = make_uint2(al[0], al[1]);
    }
}

// lv_color_fill: a lane stores one 16-byte-aligned word of 8 pixels (write-once stores); the words at the two ends of
// a buffer that does not start or end on a 16-byte boundary are stored pixel by pixel, inside the buffer only.
// n_words counts the aligned words the buffer touches, first_px = pixels of the first word that lie before buf.
static __global__ __launch_bounds__(256) void k_fill16(uint16_t * word0, uint32_t first_px, uint16_t color, size_t n, size_t n_words)
{
    const uint32_t c2 = (uint32_t)color * 0x00010001u;
    const uint4 v = make_uint4(c2, c2, c2, c2);
    for(size_t w = xcd_remap(blockIdx.x, gridDim.x) * (size_t)256 + threadIdx.x; w < n_words; w += (size_t)gridDim.x * 256) {
        const long long lo = (long long)w * 8 - first_px;        // index in buf of the word's first pixel
        uint16_t * p = word0 + w * 8;
        if(lo >= 0 && (size_t)lo + 8 <= n) store_write_once(p, v);
        else {
            const int j0 = lo < 0 ? (int)-lo : 0, j1 = (long long)n - lo < 8 ? (int)((long long)n - lo) : 8;
            store_group_masked(p, v, ((1u << j1) - 1u) & ~((1u << j0) - 1u));
        }
    }
}
__device__ __forceinline__ uint32_t ui_alpha(uint64_t h)
{
    uint32_t t = (uint32_t)(h >> 56) % 3u;
    if(t == 0) return 0;
    if(t == 1) return 255;
    return 1u + ((uint32_t)(h >> 32) & 0xFFFFu) % 254u;
}
static __global__ void k_synth(void * out, int kind, uint64_t seed, uint64_t first, size_t n)
{
    for(size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        uint64_t h = fin64(seed + (first + i + 1) * 0x9E3779B97F4A7C15ULL);
        switch(kind) {
            case 0: reinterpret_cast<uint16_t *>(out)[i] = (uint16_t)(h & 0xFFFF); break;
            case 1: reinterpret_cast<uint8_t *>(out)[i] = (uint8_t)ui_alpha(h); break;
            case 2: reinterpret_cast<uint8_t *>(out)[i] = (uint8_t)(h & 0xFF); break;
            case 3: reinterpret_cast<uint32_t *>(out)[i] = (uint32_t)(h & 0xFFFFFF) | (ui_alpha(h) << 24); break;
            default: reinterpret_cast<uint32_t *>(out)[i] = (uint32_t)h; break;
        }
    }
}
static __global__ void k_checksum(const void * in, int elem_bytes, uint64_t first, size_t n, unsigned long long * sum)
{
    uint64_t s = 0;
    for(size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        uint64_t v = elem_bytes == 1   ? reinterpret_cast<const uint8_t *>(in)[i]
                     : elem_bytes == 2 ? reinterpret_cast<const uint16_t *>(in)[i]
                                       : reinterpret_cast<const uint32_t *>(in)[i];
        s += fin64((first + i) * 0x9E3779B97F4A7C15ULL + v * 0xD1B54A32D192ED03ULL + 1);
    }
#pragma unroll
    for(int sft = 32; sft > 0; sft >>= 1) {
        uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)s, sft, 64);
        uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(s >> 32), sft, 64);
        s += ((uint64_t)hi << 32) | lo;
    }
    if((threadIdx.x & 63) == 0) atomicAdd(sum, (unsigned long long)s);
}

// one count per workgroup, binned by the XCD it ran on (HW_REG_XCC_ID, bits 3:0)
static __global__ void k_xcc_census(uint32_t * counts8)
{
    if(threadIdx.x == 0) {
        uint32_t id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        atomicAdd(&counts8[id & 7u], 1u);
    }
    // give the dispatcher time to spread workgroups over every CU the queue may use
    for(int i = 0; i < 200; i++) __builtin_amdgcn_s_sleep(20);
}

}  // namespace lvgpu
