// lvgpu_px.hpp -- per-pixel arithmetic of LVGL's software renderer as pure device functions.
//
// Every function works on "native" RGB565 values (R in bits 11-15, G 5-10, B 0-4).  Buffers stored with
// LV_COLOR_16_SWAP 1 hold the same value with its two bytes exchanged (libraries/lvgl/src/misc/lv_color.h:208-222),
// so loads/stores go through px_swap16x2 when swap is set and the arithmetic itself is byte-order free; this is what
// lv_color_mix does internally too (lv_color.h:443-446, 454-456).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lvgpu {

enum SrcFmt : int { SRC_FILL = 0, SRC_RGB565 = 1, SRC_ARGB8888 = 2, SRC_RGBA8888 = 3, SRC_ARGB8565 = 4, SRC_COPY = 5 };

// How the effective opacity of one pixel is derived (libraries/lvgl/src/draw/sw/lv_draw_sw_blend.c):
enum OpaMode : int {
    OM_CONST = 0,         // fill_normal :254-259 / map_normal :655-670 (no mask): o = opa (255 when opa >= LV_OPA_MAX)
    OM_MASK = 1,          // fill_normal :295-344 / map_normal :675-718: o = mask
    OM_FILL_MASK_OPA = 2, // fill_normal :346-373: mask==255 ? opa : (mask*opa)>>8
    OM_MAP_MASK_OPA = 3,  // map_normal :720-732 and fill_/map_blended: mask>=253 ? opa : (opa*mask)>>8
    OM_FILL_PREMULT = 4   // fill_normal :261-287: premultiplied formula with the first-pixel cache
};

enum BlendFn : int { BF_NORMAL = 0, BF_ADDITIVE = 1, BF_SUBTRACTIVE = 2, BF_MULTIPLY = 3 };

__device__ __forceinline__ uint32_t px_swap16x2(uint32_t v)  // exchange the bytes of both 16-bit halves
{
    return __builtin_amdgcn_perm(v, v, 0x02030001u);
}
// The same under a run-time flag, as one instruction either way: the byte selector is the identity when swap == 0
// (a branch around the exchange gets if-converted into the exchange plus a select per dword).
__device__ __forceinline__ uint32_t px_swap16x2_if(uint32_t v, int swap)
{
    return __builtin_amdgcn_perm(v, v, swap ? 0x02030001u : 0x03020100u);
}
__device__ __forceinline__ uint32_t px_swap16(uint32_t v)    // v < 65536
{
    return ((v << 8) | (v >> 8)) & 0xFFFFu;
}

// lv_color_mix, libraries/lvgl/src/misc/lv_color.h:447-453, on native values, with mix already reduced to
// m5 = (mix + 4) >> 3 (0..32).  uint32 modular arithmetic exactly as in the reference.
__device__ __forceinline__ uint32_t px_mix5(uint32_t fg, uint32_t bg, uint32_t m5)
{
    uint32_t b = (bg | (bg << 16)) & 0x07E0F81Fu;
    uint32_t f = (fg | (fg << 16)) & 0x07E0F81Fu;
    uint32_t r = ((((f - b) * m5) >> 5) + b) & 0x07E0F81Fu;
    return ((r >> 16) | r) & 0xFFFFu;
}
__device__ __forceinline__ uint32_t px_mix(uint32_t fg, uint32_t bg, uint32_t mix)
{
    return px_mix5(fg, bg, (mix + 4u) >> 3);
}

// LV_COLOR_MAKE16 (lv_color.h:148-154) from a little-endian lv_color32_t word 0xAARRGGBB (lv_color.h:224-232).
__device__ __forceinline__ uint32_t px_from_argb8888(uint32_t c)
{
    return ((c >> 8) & 0xF800u) | ((c >> 5) & 0x07E0u) | ((c >> 3) & 0x001Fu);
}
// ... and from lodepng's byte order R,G,B,A = little-endian word 0xAABBGGRR.
__device__ __forceinline__ uint32_t px_from_rgba8888(uint32_t c)
{
    return ((c << 8) & 0xF800u) | ((c >> 5) & 0x07E0u) | ((c >> 19) & 0x001Fu);
}

// lv_color_to32, lv_color.h:418-423.
__device__ __forceinline__ uint32_t px_to32(uint32_t c)
{
    uint32_t r = ((c >> 11) * 263u + 7u) >> 5;
    uint32_t g = (((c >> 5) & 0x3Fu) * 259u + 3u) >> 6;
    uint32_t b = ((c & 0x1Fu) * 263u + 7u) >> 5;
    return 0xFF000000u | (r << 16) | (g << 8) | b;
}

// color_blend_true_color_additive / _subtractive / _multiply, lv_draw_sw_blend.c:950-1050, the part before the
// final lv_color_mix.  swap selects the LV_COLOR_16_SWAP 1 build, whose subtractive green ADDS (:1015-1018) and whose
// multiply green is affected by a double macro evaluation (:1040).
template <int BF>
__device__ __forceinline__ uint32_t px_blend_fn(uint32_t fg, uint32_t bg, int swap)
{
    if(BF == BF_NORMAL) return fg;
    uint32_t fr = fg >> 11, fgn = (fg >> 5) & 0x3Fu, fb = fg & 0x1Fu;
    uint32_t br = bg >> 11, bgn = (bg >> 5) & 0x3Fu, bb = bg & 0x1Fu;
    uint32_t r, g, b;
    if(BF == BF_ADDITIVE) {
        r = min(br + fr, 31u);
        g = min(bgn + fgn, 63u);
        b = min(bb + fb, 31u);
    }
    else if(BF == BF_SUBTRACTIVE) {
        r = br > fr ? br - fr : 0u;
        b = bb > fb ? bb - fb : 0u;
        g = swap ? ((bgn + fgn) & 0x3Fu) : (bgn > fgn ? bgn - fgn : 0u);
    }
    else {
        r = (fr * br) >> 5;
        b = (fb * bb) >> 5;
        g = (fgn * bgn) >> 6;
        if(swap) {
            // LV_COLOR_SET_G16 of the swapped build (lv_color.h:134) evaluates its argument twice; the second
            // evaluation (green_l) sees the foreground green with the freshly stored high bits (:1040).
            const uint32_t gh = (g >> 3) & 7u;
            const uint32_t g2 = (((gh << 3) + (fgn & 7u)) * bgn) >> 6;
            g = (gh << 3) | (g2 & 7u);
        }
    }
    return (r << 11) | (g << 5) | b;
}

// A display with anti-aliasing off (disp->driver->antialiasing == 0): lv_draw_sw_blend_basic rounds every mask value
// to 0 or 255 before it blends, `mask[i] > 128 ? LV_OPA_COVER : LV_OPA_TRANSP` (lv_draw_sw_blend.c:150-156).
__device__ __forceinline__ uint32_t px_mask_no_aa(uint32_t m) { return m > 128u ? 255u : 0u; }
// ... four mask bytes at once: a byte is above 128 iff its top bit is set and any of its other seven is.
__device__ __forceinline__ uint32_t px_mask_no_aa_x4(uint32_t w)
{
    const uint32_t above = ((w & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) & w & 0x80808080u;
    return (above >> 7) * 255u;
}

// Effective opacity per pixel for the masked modes.
template <int OM>
__device__ __forceinline__ uint32_t px_opa(uint32_t m, uint32_t opa)
{
    if(OM == OM_MASK) return m;
    if(OM == OM_FILL_MASK_OPA) return m == 255u ? opa : (m * opa) >> 8;  // m == 0 gives 0 -> destination unchanged
    if(OM == OM_MAP_MASK_OPA) return m == 0u ? 0u : (m >= 253u ? opa : (opa * m) >> 8);
    return opa;
}

// The 5-bit ratio lv_color_mix works with, (o + 4) >> 3, straight from the mask and the opacity for the two
// mask-times-opacity modes: ((m * opa >> 8) + 4) >> 3 = (m * opa + 1024) >> 11 (both divisions round down), one
// multiply-add and one shift; the mask values that take the opacity as it is select the launch-constant (opa + 4) >> 3.
template <int OM>
__device__ __forceinline__ uint32_t px_mix5_of(uint32_t m, uint32_t opa)
{
    if(OM == OM_FILL_MASK_OPA) return m == 255u ? (opa + 4u) >> 3 : (__umul24(m, opa) + 1024u) >> 11;   // v_mad_u32_u24
    if(OM == OM_MAP_MASK_OPA) return m >= 253u ? (opa + 4u) >> 3 : (__umul24(m, opa) + 1024u) >> 11;   // m == 0 gives 0
    return ((OM == OM_MASK ? m : opa) + 4u) >> 3;
}

// Constants of fill_normal's no-mask translucent branch (lv_draw_sw_blend.c:261-287), prepared on the host.
struct PremultFill {
    uint32_t pm_r, pm_g, pm_b;  // lv_color_premult(color, opa_q)           (lv_color.h:474-488)
    uint32_t opa_inv;           // 255 - opa_q, opa_q = ((opa+4)>>3)<<3 in uint8 (:269-275)
    uint32_t first_res;         // native lv_color_mix(color, black, opa): what the initial cache entry holds (:262-263)
};

// lv_color_mix_premult (lv_color.h:499-519) with LV_UDIV255 (lv_math.h:132) on a native destination value.
__device__ __forceinline__ uint32_t px_mix_premult(const PremultFill & pf, uint32_t d)
{
    uint32_t r = ((pf.pm_r + (d >> 11) * pf.opa_inv) * 0x8081u) >> 23;
    uint32_t g = ((pf.pm_g + ((d >> 5) & 0x3Fu) * pf.opa_inv) * 0x8081u) >> 23;
    uint32_t b = ((pf.pm_b + (d & 0x1Fu) * pf.opa_inv) * 0x8081u) >> 23;
    return ((r & 0x1Fu) << 11) | ((g & 0x3Fu) << 5) | (b & 0x1Fu);
}

// The same for the two native pixels held in the halves of w, in packed 16-bit arithmetic: every sum
// t = pm + channel * opa_inv stays below 2^16 (opa_q + opa_inv = 255, so it is at most 63 * 255), and for such t
// LV_UDIV255(t) = (t * 0x8081) >> 23 = floor(t / 255) = (t1 + (t1 >> 8)) >> 8 with t1 = t + 1 (checked for every t up to
// 63 * 255), which needs no 32-bit product; the + 1 rides on the premultiplied constant.  A quotient is at most the
// channel's maximum, so the reference's masks on the results (:514-516) change nothing and are left out.
typedef unsigned short px_u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t px_mix_premult_x2(const PremultFill & pf, uint32_t w)
{
    auto pk = [](uint32_t v) { return __builtin_bit_cast(px_u16x2, v); };
    auto u32 = [](px_u16x2 v) { return __builtin_bit_cast(uint32_t, v); };
    const px_u16x2 inv = pk(pf.opa_inv * 0x00010001u);
    auto div255 = [&](px_u16x2 t1) { return (px_u16x2)((t1 + (t1 >> 8)) >> 8); };
    const px_u16x2 r = div255(pk((w >> 11) & 0x001F001Fu) * inv + pk((pf.pm_r + 1u) * 0x00010001u));
    const px_u16x2 g = div255(pk((w >> 5) & 0x003F003Fu) * inv + pk((pf.pm_g + 1u) * 0x00010001u));
    const px_u16x2 b = div255(pk(w & 0x001F001Fu) * inv + pk((pf.pm_b + 1u) * 0x00010001u));
    return (u32(r) << 11) | (u32(g) << 5) | u32(b);
}

// The scan-order cache of :262-263/:279-283 in closed form: a black destination pixel that precedes the first
// non-black pixel of the area (row-major scan order) still sees the initial cache entry; every other pixel is
// lv_color_mix_premult of its own destination value.
__device__ __forceinline__ uint32_t px_fill_premult(const PremultFill & pf, uint32_t d, uint32_t scan_idx,
                                                    uint32_t first_nonblack)
{
    if(d == 0u && scan_idx < first_nonblack) return pf.first_res;
    return px_mix_premult(pf, d);
}

// finaliser of splitmix64: the counter-based generator behind lvgpu_synth / lvgpu_checksum
__device__ __forceinline__ uint64_t fin64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

}  // namespace lvgpu
