// lvgpu_list.hpp -- display-list kernel: an ordered list of blend descriptors applied to a batch of draw buffers in
// ONE launch.  LVGL's producers call lv_draw_sw_blend once per object part or even once per row
// (libraries/lvgl/src/draw/sw/lv_draw_sw_rect.c:118-330, lv_draw_sw_letter.c:246-398, lv_draw_sw_line.c,
// lv_draw_sw_arc.c); a scene is hundreds to thousands of small descriptors.  Launching one kernel per descriptor
// re-reads and re-writes the draw buffers once per layer and is launch-bound for the small ones.  Here the buffer
// is cut into 64x64-pixel tiles, the host bins the descriptors by tile, and a wavefront keeps its tile in LDS while
// it applies the tile's descriptors in list order: the draw buffer is read once and written once per list whatever
// the depth of the scene, and the result is what the same descriptors give when applied one after the other
// (pixels are independent within a descriptor and every pixel sees the descriptors in list order).
#pragma once
#include "lvgpu_kernels.hpp"

namespace lvgpu {

struct ListItem {
    const void * src;      // first source pixel of the clipped area in frame 0
    const uint8_t * mask;  // first mask byte of the clipped area in frame 0, or nullptr
    long long src_fs;      // source pixels between frames
    long long mask_fs;     // bytes between frames
    int src_stride;        // pixels
    int mask_stride;       // bytes
    int x1, y1, x2, y2;    // clipped area, pixels relative to the buffer origin, inclusive
    uint32_t color;        // native fill colour
    uint32_t opa;
    int opa_mode;          // OpaMode
    int blend_fn;          // BlendFn
    int fmt;               // source format
    int swap;              // LV_COLOR_16_SWAP
    PremultFill pf;        // OM_FILL_PREMULT only (lists hold such fills only when black needs no scan-order cache)
    int raw_mode;          // ARGB8565 destination: dsc->blend_mode as given (opa then holds dsc->opa as given)
    int has_mask;          // ARGB8565 destination: a per-pixel alpha exists (mask buffer or source alpha)
};

struct ListArgs {
    uint16_t * dst;
    long long dst_fs;
    int W, H;                    // buffer size, W % 8 == 0
    uint32_t n_frames;
    const ListItem * items;
    const uint32_t * bin_start;  // per tile: first step of the tile in steps; tiles_x*tiles_y + 1 entries
    const uint4 * steps;         // what a wave does to its tile, in list order (ListStep)
    const uint4 * chunks;        // 8-pixel pieces of small items, grouped into batches (ListChunk)
    const uint32_t * sel;        // tiles to process: tile index, bit 31 set = the tile's first item overwrites it whole
    uint32_t n_sel;
    uint32_t tiles_x;
    int swap;
    uint16_t * out;              // rotated / copied panel image (frame 0) or nullptr
    long long out_fs;
    int rot;
    int swap_out;
};

constexpr int LT = 64;  // tile edge, pixels

// Wavefronts that share one tile (WPT; NT = 64 * WPT threads).  With two, the tile's LDS (8 KB) is divided among
// more waves, so more waves fit a CU (the LDS, not the registers, limits occupancy) and a tile's serial chain of
// steps gets shorter; the price is a block barrier instead of a wave-level fence between steps, and idle lanes when
// the steps are small.  The launcher picks WPT from the average batch size of the list.
template <int WPT>
struct ListCfg {
    static constexpr int NT = 64 * WPT;                    // threads per tile
    static constexpr int BLOCK = WPT == 1 ? 256 : NT;      // threads per block
    static constexpr int TPB = BLOCK / NT;                 // tiles per block
};
template <int WPT>
__device__ __forceinline__ void tile_fence()
{
    if(WPT == 1) wave_lds_fence();
    else __syncthreads();
}

// A step is either one item applied to its whole region of the tile (x = LIST_STEP_ITEM | item index), or a batch of
// chunks (x = kind, y = first chunk, z = number of whole chunks, w = number of partial chunks, whole ones first).
// A chunk is one 16-byte group of one tile row touched by one small item: x = item index, y = row | group << 6 |
// pixel mask << 9, z / w = where the group's first pixel lies in the item's mask (bytes) / source (pixels), relative
// to the clipped area's first element (signed: a partial group may start left of the area).  The chunks of a batch come from consecutive items of the same kind (source format, opacity mode,
// blend function, mask or no mask) that touch pairwise different groups, so they can be applied side by side by the
// lanes of the wave; that is what turns LVGL's hundreds of one-row descriptors per object into a few wave passes.
constexpr uint32_t LIST_STEP_ITEM = 0x80000000u;
// kind = source format | opacity mode (OM_* or OM_OPAQUE; 7 = decide per pixel) << 3 | (blend_fn != normal) << 6
__host__ __device__ inline uint32_t list_kind(int fmt, int om, bool blended) { return (uint32_t)fmt | ((uint32_t)om << 3) | (blended ? 64u : 0u); }

// Position of pixel (row, col) of the tile in the wave's LDS area, in pixels.  Rows are 128 bytes; the 16-byte
// chunks of a row are XOR-swizzled with the row's 8-row band so that the column gathers of the 90/270-degree output
// (8 lanes reading the same column of 8 different bands) fall into different banks.
__device__ __forceinline__ int lt_pos(int row, int col)
{
    return row * LT + ((((col >> 3) ^ (row >> 3)) & 7) << 3) + (col & 7);
}

// Source colour and alpha of one pixel at area-relative (ax, ay).
template <int SRC>
__device__ __forceinline__ void list_fetch(const ListItem & it, uint32_t f, int ax, int ay, uint32_t & fg, uint32_t & m)
{
    const long long sp = (long long)f * it.src_fs + (long long)ay * it.src_stride + ax;
    if(SRC == SRC_RGB565) {
        const uint32_t v = reinterpret_cast<const uint16_t *>(it.src)[sp];
        fg = it.swap ? px_swap16(v) : v;
    }
    else if(SRC == SRC_ARGB8888 || SRC == SRC_RGBA8888) {
        const uint32_t v = reinterpret_cast<const uint32_t *>(it.src)[sp];
        fg = SRC == SRC_ARGB8888 ? px_from_argb8888(v) : px_from_rgba8888(v);
        m = v >> 24;
    }
    else if(SRC == SRC_ARGB8565) {
        const uint8_t * p = reinterpret_cast<const uint8_t *>(it.src) + sp * 3;
        const uint32_t v = p[0] | (p[1] << 8);
        fg = it.swap ? px_swap16(v) : v;
        m = p[2];
    }
    if(it.mask) m = it.mask[(long long)f * it.mask_fs + (long long)ay * it.mask_stride + ax];
}

// The same with the element positions given: si = source pixel index, mi = mask byte index (both include the frame).
template <int SRC>
__device__ __forceinline__ void list_fetch_at(const ListItem & it, int swap, long long si, long long mi, uint32_t & fg, uint32_t & m)
{
    if(SRC == SRC_RGB565) {
        const uint32_t v = reinterpret_cast<const uint16_t *>(it.src)[si];
        fg = swap ? px_swap16(v) : v;
    }
    else if(SRC == SRC_ARGB8888 || SRC == SRC_RGBA8888) {
        const uint32_t v = reinterpret_cast<const uint32_t *>(it.src)[si];
        fg = SRC == SRC_ARGB8888 ? px_from_argb8888(v) : px_from_rgba8888(v);
        m = v >> 24;
    }
    else if(SRC == SRC_ARGB8565) {
        const uint8_t * p = reinterpret_cast<const uint8_t *>(it.src) + si * 3;
        const uint32_t v = p[0] | (p[1] << 8);
        fg = swap ? px_swap16(v) : v;
        m = p[2];
    }
    if(it.mask) m = it.mask[mi];
}

// The part of an item that falls into the wave's tile, in tile-relative pixels; (ox, oy) turns a tile-relative
// position into a position relative to the item's clipped area.
struct ListRegion {
    int x1, y1, x2, y2;
    int ox, oy;
};

// idx / d for idx < 4096, 1 <= d <= 64, exact: (idx + 0.5) / d stays at least 1/128 away from an integer
__device__ __forceinline__ uint32_t small_div(uint32_t idx, float inv_d)
{
    return (uint32_t)(((float)idx + 0.5f) * inv_d);
}

// Columns [x1, a_end) and [b_start, x2] of the region's rows -- what is left and right of the whole 8-pixel groups,
// or the whole region when it holds no whole group (a_end = x2 + 1) -- one pixel per lane and step.
constexpr int LP_UNROLL = 4;
template <int SRC, int OM, bool BLENDED, int LNT>
__device__ __forceinline__ void list_pixels(const ListItem & it, uint16_t * tile, uint32_t f, const ListRegion & r, int a_end,
                                            int b_start, int lane)
{
    const uint32_t wa = (uint32_t)(a_end - r.x1), wb = (uint32_t)(r.x2 + 1 - b_start);
    const uint32_t rw = wa + wb, n = rw * (uint32_t)(r.y2 - r.y1 + 1);
    if(n == 0) return;
    const float inv = 1.0f / (float)rw;
    for(uint32_t base = (uint32_t)lane; base < n; base += LNT * LP_UNROLL) {
        uint32_t d[LP_UNROLL], fg[LP_UNROLL], m[LP_UNROLL];
        int li[LP_UNROLL];
#pragma unroll
        for(int u = 0; u < LP_UNROLL; u++) {
            const uint32_t idx = base + (uint32_t)LNT * u;
            if(idx < n) {
                const uint32_t ly = small_div(idx, inv), lx = idx - ly * rw;
                const int col = lx < wa ? r.x1 + (int)lx : b_start + (int)(lx - wa), row = r.y1 + (int)ly;
                li[u] = lt_pos(row, col);
                fg[u] = it.color;
                m[u] = 0;
                list_fetch<SRC>(it, f, col + r.ox, row + r.oy, fg[u], m[u]);
                d[u] = OM == OM_OPAQUE ? 0u : tile[li[u]];
            }
        }
#pragma unroll
        for(int u = 0; u < LP_UNROLL; u++) {
            if(base + (uint32_t)LNT * u < n) {
                uint32_t dn = it.swap ? px_swap16(d[u]) : d[u];
                uint32_t res = px_core_s<OM, BLENDED>(it, dn, fg[u], m[u], 0u, 0u);
                if(it.swap) res = px_swap16(res);
                tile[li[u]] = (uint16_t)res;
            }
        }
    }
}

// Group columns g_lo..g_hi-1 (8 pixels each) of the region's rows, one group per lane and step.
constexpr int LG_UNROLL = 2;
template <int SRC, int OM, bool BLENDED, int LNT>
__device__ __forceinline__ void list_groups(const ListItem & it, uint4 * tile4, uint32_t f, const ListRegion & r, int g_lo,
                                            int g_hi, int lane)
{
    const uint32_t ng = (uint32_t)(g_hi - g_lo), n = ng * (uint32_t)(r.y2 - r.y1 + 1);
    const float inv = 1.0f / (float)ng;
    for(uint32_t base = (uint32_t)lane; base < n; base += LNT * LG_UNROLL) {
        uint4 d[LG_UNROLL];
        uint2 mk[LG_UNROLL];
        uint32_t sw[LG_UNROLL][SrcWords<SRC>::N];
        int li[LG_UNROLL];
#pragma unroll
        for(int u = 0; u < LG_UNROLL; u++) {
            const uint32_t idx = base + (uint32_t)LNT * u;
            mk[u] = make_uint2(0, 0);
            if(idx < n) {
                const uint32_t ly = small_div(idx, inv), lg = idx - ly * ng;
                const int row = r.y1 + (int)ly, g = g_lo + (int)lg;
                li[u] = row * 8 + ((g ^ (row >> 3)) & 7);
                const int ax = g * 8 + r.ox, ay = row + r.oy;
                if(SRC != SRC_FILL)
                    load_src_unaligned<SRC>(it.src, (long long)f * it.src_fs + (long long)ay * it.src_stride + ax, sw[u]);
                if(it.mask) {
                    u32x2_a1 v = *reinterpret_cast<const u32x2_a1 *>(it.mask + (long long)f * it.mask_fs +
                                                                     (long long)ay * it.mask_stride + ax);
                    mk[u] = make_uint2(v.x, v.y);
                }
                d[u] = OM == OM_OPAQUE ? make_uint4(0, 0, 0, 0) : tile4[li[u]];
            }
        }
#pragma unroll
        for(int u = 0; u < LG_UNROLL; u++) {
            if(base + (uint32_t)LNT * u < n) tile4[li[u]] = blend_group_s<SRC, OM, BLENDED, false>(it, d[u], sw[u], mk[u], 0u, 0u);
        }
    }
}

// One item on one tile: whole 8-pixel groups of the tile that lie inside the region go as vectors, the columns left
// and right of them (and regions narrower than a group) pixel by pixel.
template <int SRC, int OM, bool BLENDED, int LNT>
__device__ __forceinline__ void list_item(const ListItem & it, uint4 * tile4, uint32_t f, const ListRegion r, int lane)
{
    const int g_lo = (r.x1 + 7) >> 3, g_hi = (r.x2 + 1) >> 3;
    uint16_t * tile = reinterpret_cast<uint16_t *>(tile4);
    if(g_hi > g_lo) {
        list_groups<SRC, OM, BLENDED, LNT>(it, tile4, f, r, g_lo, g_hi, lane);
        list_pixels<SRC, OM, BLENDED, LNT>(it, tile, f, r, g_lo * 8, g_hi * 8, lane);
    }
    else list_pixels<SRC, OM, BLENDED, LNT>(it, tile, f, r, r.x2 + 1, r.x2 + 1, lane);
}

// Pick the specialisation for the item's (uniform) opacity mode and blend function.  The combinations the host
// produces are listed; anything else takes the run-time path.
template <int SRC, int LNT>
__device__ __forceinline__ void list_dispatch(const ListItem & it, uint4 * tile4, uint32_t f, const ListRegion & r, int lane)
{
    constexpr bool own_alpha = SRC == SRC_ARGB8888 || SRC == SRC_RGBA8888 || SRC == SRC_ARGB8565;
    const bool normal = it.blend_fn == BF_NORMAL;
    if(normal) {
        switch(it.opa_mode) {
            case OM_CONST:
                if(!own_alpha && it.opa == 255u) return list_item<SRC, OM_OPAQUE, false, LNT>(it, tile4, f, r, lane);
                if(!own_alpha) return list_item<SRC, OM_CONST, false, LNT>(it, tile4, f, r, lane);
                break;
            case OM_MASK: return list_item<SRC, OM_MASK, false, LNT>(it, tile4, f, r, lane);
            case OM_FILL_MASK_OPA:
                if(SRC == SRC_FILL) return list_item<SRC, OM_FILL_MASK_OPA, false, LNT>(it, tile4, f, r, lane);
                break;
            case OM_MAP_MASK_OPA:
                if(SRC != SRC_FILL) return list_item<SRC, OM_MAP_MASK_OPA, false, LNT>(it, tile4, f, r, lane);
                break;
            case OM_FILL_PREMULT:
                if(SRC == SRC_FILL) return list_item<SRC, OM_FILL_PREMULT, false, LNT>(it, tile4, f, r, lane);
                break;
            default: break;
        }
    }
    list_item<SRC, OM_RUNTIME, true, LNT>(it, tile4, f, r, lane);
}

// What blend_group_s / px_core_s read, per lane (the lanes of a batch may work for different items).
struct LaneView {
    uint32_t color, opa;
    int opa_mode, blend_fn, swap;
    PremultFill pf;
};

template <int SRC, int OM, bool BLENDED, int LNT>
__device__ __forceinline__ void list_batch(const ListArgs & a, uint4 * tile4, uint32_t f, int tx, int ty, uint32_t first,
                                           uint32_t n_full, uint32_t n_part, int lane)
{
    const uint4 * ch = a.chunks + first;
    const uint32_t n = n_full + n_part;
    uint4 c_next = lane < (int)n ? ch[lane] : make_uint4(0, 0, 0, 0);
    for(uint32_t base = 0; base < n; base += LNT) {
        const uint32_t idx = base + (uint32_t)lane;
        const uint4 c = c_next;
        if(idx + LNT < n) c_next = ch[idx + LNT];  // requested now, used in the next pass
        if(idx >= n) continue;
        const ListItem & it = a.items[c.x];
        const int row = (int)(c.y & 63u), g = (int)((c.y >> 6) & 7u);
        const uint32_t pxmask = (c.y >> 9) & 0xFFu;
        const int li = row * 8 + ((g ^ (row >> 3)) & 7);
        const long long mi = (long long)f * it.mask_fs + (int)c.z, si = (long long)f * it.src_fs + (int)c.w;
        LaneView v;
        v.color = it.color;
        v.opa = it.opa;
        v.opa_mode = it.opa_mode;
        v.blend_fn = it.blend_fn;
        v.swap = a.swap;
        if(OM == OM_FILL_PREMULT || OM == OM_RUNTIME) v.pf = it.pf;
        if(idx < n_full) {  // the whole group lies inside the item: vector accesses
            uint32_t sw[SrcWords<SRC>::N];
            uint2 mk = make_uint2(0, 0);
            if(SRC != SRC_FILL) load_src_unaligned<SRC>(it.src, si, sw);
            if(it.mask) {
                u32x2_a1 m = *reinterpret_cast<const u32x2_a1 *>(it.mask + mi);
                mk = make_uint2(m.x, m.y);
            }
            const uint4 d = OM == OM_OPAQUE ? make_uint4(0, 0, 0, 0) : tile4[li];
            tile4[li] = blend_group_s<SRC, OM, BLENDED, false>(v, d, sw, mk, 0u, 0u);
        }
        else {  // only the pixels of pxmask belong to the item: nothing outside its rectangle is read
            const uint4 d = tile4[li];
            uint32_t dw[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
            for(int j = 0; j < 8; j++) {
                if((pxmask >> j) & 1u) {
                    uint32_t fg = v.color, m = 0;
                    list_fetch_at<SRC>(it, v.swap, si + j, mi + j, fg, m);
                    uint32_t dn = (dw[j >> 1] >> ((j & 1) * 16)) & 0xFFFFu;
                    if(v.swap) dn = px_swap16(dn);
                    uint32_t res = px_core_s<OM, BLENDED>(v, dn, fg, m, 0u, 0u);
                    if(v.swap) res = px_swap16(res);
                    dw[j >> 1] = (j & 1) ? (dw[j >> 1] & 0x0000FFFFu) | (res << 16) : (dw[j >> 1] & 0xFFFF0000u) | res;
                }
            }
            tile4[li] = make_uint4(dw[0], dw[1], dw[2], dw[3]);
        }
    }
}

template <int SRC, int LNT>
__device__ __forceinline__ void batch_dispatch(const ListArgs & a, uint32_t kind, uint4 * tile4, uint32_t f, int tx, int ty,
                                               uint32_t first, uint32_t n_full, uint32_t n_part, int lane)
{
    constexpr bool own_alpha = SRC == SRC_ARGB8888 || SRC == SRC_RGBA8888 || SRC == SRC_ARGB8565;
    const int om = (int)((kind >> 3) & 7u);
    if(!(kind & 64u)) {
        switch(om) {
            case OM_OPAQUE:
                if(!own_alpha) return list_batch<SRC, OM_OPAQUE, false, LNT>(a, tile4, f, tx, ty, first, n_full, n_part, lane);
                break;
            case OM_CONST:
                if(!own_alpha) return list_batch<SRC, OM_CONST, false, LNT>(a, tile4, f, tx, ty, first, n_full, n_part, lane);
                break;
            case OM_MASK: return list_batch<SRC, OM_MASK, false, LNT>(a, tile4, f, tx, ty, first, n_full, n_part, lane);
            case OM_FILL_MASK_OPA:
                if(SRC == SRC_FILL) return list_batch<SRC, OM_FILL_MASK_OPA, false, LNT>(a, tile4, f, tx, ty, first, n_full, n_part, lane);
                break;
            case OM_MAP_MASK_OPA:
                if(SRC != SRC_FILL) return list_batch<SRC, OM_MAP_MASK_OPA, false, LNT>(a, tile4, f, tx, ty, first, n_full, n_part, lane);
                break;
            case OM_FILL_PREMULT:
                if(SRC == SRC_FILL) return list_batch<SRC, OM_FILL_PREMULT, false, LNT>(a, tile4, f, tx, ty, first, n_full, n_part, lane);
                break;
            default: break;
        }
    }
    list_batch<SRC, OM_RUNTIME, true, LNT>(a, tile4, f, tx, ty, first, n_full, n_part, lane);
}

template <int WPT>
__global__ __launch_bounds__(ListCfg<WPT>::BLOCK) void k_blend_list(ListArgs a)
{
    constexpr int LNT = ListCfg<WPT>::NT, LIST_TPB = ListCfg<WPT>::TPB;
    __shared__ uint4 lds4[LIST_TPB][LT * LT / 8];
    const int lane = threadIdx.x % LNT;  // position among the threads that share the tile
    const int sub = threadIdx.x / LNT;
    const uint32_t wt = __builtin_amdgcn_readfirstlane(xcd_remap(blockIdx.x, gridDim.x) * LIST_TPB + sub);
    if(wt >= a.n_sel * a.n_frames) return;  // tiles of one block never wait for each other below
    const uint32_t f = wt / a.n_sel;
    const uint32_t sel = a.sel[wt - f * a.n_sel];
    const uint32_t t = sel & 0x7FFFFFFFu;
    const bool overwrite = (sel >> 31) != 0;
    const int ty = (int)(t / a.tiles_x), tx = (int)(t - (uint32_t)ty * a.tiles_x);
    const int W = a.W, H = a.H;
    uint16_t * dst = a.dst + (long long)f * a.dst_fs;
    uint4 * tile4 = lds4[sub];
    uint16_t * tile = reinterpret_cast<uint16_t *>(tile4);
    const uint32_t k0 = a.bin_start[t], k1 = a.bin_start[t + 1];
    constexpr int N_ACC = LT * LT / 8 / LNT;  // 16-byte accesses per thread to move the tile

    if(!overwrite) {
#pragma unroll
        for(int i = 0; i < N_ACC; i++) {
            const int q = lane + LNT * i, row = q >> 3, c = q & 7;
            const int y = ty * LT + row, gx = tx * LT + c * 8;
            if(y < H && gx < W) tile4[row * 8 + (c ^ (row >> 3))] = *reinterpret_cast<const uint4 *>(dst + (size_t)y * W + gx);
        }
    }
    tile_fence<WPT>();

    uint4 st_next = k0 < k1 ? a.steps[k0] : make_uint4(0, 0, 0, 0);
    for(uint32_t k = k0; k < k1; k++) {
        const uint32_t s_x = __builtin_amdgcn_readfirstlane(st_next.x), s_y = __builtin_amdgcn_readfirstlane(st_next.y);
        const uint32_t s_z = __builtin_amdgcn_readfirstlane(st_next.z), s_w = __builtin_amdgcn_readfirstlane(st_next.w);
        if(k + 1 < k1) st_next = a.steps[k + 1];  // requested now, needed after this step
        if(s_x & LIST_STEP_ITEM) {
            const ListItem & it = a.items[s_x & ~LIST_STEP_ITEM];
            ListRegion r;
            r.x1 = max(it.x1, tx * LT) - tx * LT;
            r.x2 = min(it.x2, tx * LT + LT - 1) - tx * LT;
            r.y1 = max(it.y1, ty * LT) - ty * LT;
            r.y2 = min(it.y2, ty * LT + LT - 1) - ty * LT;
            r.ox = tx * LT - it.x1;  // tile-relative -> area-relative
            r.oy = ty * LT - it.y1;
            if(r.x2 >= r.x1 && r.y2 >= r.y1) {
                switch(it.fmt) {
                    case SRC_FILL: list_dispatch<SRC_FILL, LNT>(it, tile4, f, r, lane); break;
                    case SRC_RGB565: list_dispatch<SRC_RGB565, LNT>(it, tile4, f, r, lane); break;
                    case SRC_ARGB8888: list_dispatch<SRC_ARGB8888, LNT>(it, tile4, f, r, lane); break;
                    case SRC_RGBA8888: list_dispatch<SRC_RGBA8888, LNT>(it, tile4, f, r, lane); break;
                    default: list_dispatch<SRC_ARGB8565, LNT>(it, tile4, f, r, lane); break;
                }
            }
        }
        else {
            switch(s_x & 7u) {
                case SRC_FILL: batch_dispatch<SRC_FILL, LNT>(a, s_x, tile4, f, tx, ty, s_y, s_z, s_w, lane); break;
                case SRC_RGB565: batch_dispatch<SRC_RGB565, LNT>(a, s_x, tile4, f, tx, ty, s_y, s_z, s_w, lane); break;
                case SRC_ARGB8888: batch_dispatch<SRC_ARGB8888, LNT>(a, s_x, tile4, f, tx, ty, s_y, s_z, s_w, lane); break;
                case SRC_RGBA8888: batch_dispatch<SRC_RGBA8888, LNT>(a, s_x, tile4, f, tx, ty, s_y, s_z, s_w, lane); break;
                default: batch_dispatch<SRC_ARGB8565, LNT>(a, s_x, tile4, f, tx, ty, s_y, s_z, s_w, lane); break;
            }
        }
        tile_fence<WPT>();
    }

    {
        const bool wr_dst = k1 > k0;
        uint16_t * out = a.out ? a.out + (long long)f * a.out_fs : nullptr;
        const bool linear_out = out && (a.rot == 0 || a.rot == 2);
        const size_t area = (size_t)W * H;
#pragma unroll
        for(int i = 0; i < N_ACC; i++) {
            const int q = lane + LNT * i, row = q >> 3, c = q & 7;
            const int y = ty * LT + row, gx = tx * LT + c * 8;
            if(y < H && gx < W) {
                uint4 v = tile4[row * 8 + (c ^ (row >> 3))];
                const size_t p = (size_t)y * W + gx;
                if(wr_dst) *reinterpret_cast<uint4 *>(dst + p) = v;
                if(linear_out) {
                    if(a.swap_out) v = swap_group(v);
                    if(a.rot == 2) store_streaming(out + (area - 8 - p), reverse_group(v));
                    else store_streaming(out + p, v);
                }
            }
        }
    }
    if(a.out && (a.rot == 1 || a.rot == 3)) {  // H % 8 == 0 (checked by the launcher)
        uint16_t * out = a.out + (long long)f * a.out_fs;
#pragma unroll
        for(int k = 0; k < N_ACC; k++) {
            const int q = lane + LNT * k, xl = q >> 3, c = q & 7;
            const int yb = ty * LT + c * 8;
            const int X = tx * LT + xl;
            if(X < W && yb < H) {
                uint32_t px[8];
#pragma unroll
                for(int q = 0; q < 8; q++) px[q] = tile[lt_pos(c * 8 + q, xl)];
                uint4 v;
                size_t o;
                if(a.rot == 1) {
                    v = make_uint4(px[7] | (px[6] << 16), px[5] | (px[4] << 16), px[3] | (px[2] << 16), px[1] | (px[0] << 16));
                    o = (size_t)X * H + (size_t)(H - 8 - yb);
                }
                else {
                    v = make_uint4(px[0] | (px[1] << 16), px[2] | (px[3] << 16), px[4] | (px[5] << 16), px[6] | (px[7] << 16));
                    o = (size_t)(W - 1 - X) * H + (size_t)yb;
                }
                if(a.swap_out) v = swap_group(v);
                store_streaming(out + o, v);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The same for an ARGB8565 destination (3 bytes per pixel; the draw buffer of a transparent screen or of a layer
// with alpha, lv_draw_sw_blend.c:129-133, 177-184).  The tile is 64x64x3 bytes of LDS per wave, every step goes
// pixel by pixel through argb_dst_px, there is no fused flush (the reference flushes lv_color_t buffers only).
// Requires W % 16 == 0 so that tile rows start on 16-byte boundaries.
// ---------------------------------------------------------------------------------------------------------------
struct ArgbLaneView {
    int raw_opa, raw_mode, has_mask, swap;
};

template <int SRC>
__device__ __forceinline__ void argb_list_px(const ListItem & it, int swap, uint8_t * tile, uint32_t f, int row, int col, int ax,
                                             int ay)
{
    uint32_t fg = it.color, m = 0;
    list_fetch<SRC>(it, f, ax, ay, fg, m);
    uint8_t * dp = tile + (row * LT + col) * 3;
    uint32_t bg = dp[0] | (dp[1] << 8);
    if(swap) bg = px_swap16(bg);
    ArgbLaneView v;
    v.raw_opa = (int)it.opa;
    v.raw_mode = it.raw_mode;
    v.has_mask = it.has_mask;
    v.swap = swap;
    const ArgbOut r = argb_dst_px(v, SRC == SRC_FILL, fg, m, bg, dp[2]);
    if(r.wo) dp[2] = (uint8_t)r.opa;
    if(r.wc) {
        const uint32_t c = swap ? px_swap16(r.color) : r.color;
        dp[0] = (uint8_t)(c & 0xFFu);
        dp[1] = (uint8_t)(c >> 8);
    }
}

// One whole 8-pixel group (24 bytes of the tile) of one item: vector accesses.
template <int SRC>
__device__ __forceinline__ void argb_list_group(const ListItem & it, int swap, uint8_t * tile, uint32_t f, int row, int g, int ax,
                                                int ay)
{
    uint2 * p = reinterpret_cast<uint2 *>(tile + row * (LT * 3) + g * 24);
    const uint2 v0 = p[0], v1 = p[1], v2 = p[2];
    uint32_t dw[6] = {v0.x, v0.y, v1.x, v1.y, v2.x, v2.y};
    uint32_t sw[SrcWords<SRC>::N];
    uint2 mk = make_uint2(0, 0);
    if(SRC != SRC_FILL) load_src_unaligned<SRC>(it.src, (long long)f * it.src_fs + (long long)ay * it.src_stride + ax, sw);
    if(it.mask) {
        u32x2_a1 m = *reinterpret_cast<const u32x2_a1 *>(it.mask + (long long)f * it.mask_fs + (long long)ay * it.mask_stride + ax);
        mk = make_uint2(m.x, m.y);
    }
    ArgbLaneView v;
    v.raw_opa = (int)it.opa;
    v.raw_mode = it.raw_mode;
    v.has_mask = it.has_mask;
    v.swap = swap;
    argb_group<SRC>(v, dw, sw, mk, it.color);
    p[0] = make_uint2(dw[0], dw[1]);
    p[1] = make_uint2(dw[2], dw[3]);
    p[2] = make_uint2(dw[4], dw[5]);
}

template <int SRC>
__device__ __forceinline__ void argb_list_item(const ListItem & it, int swap, uint8_t * tile, uint32_t f, const ListRegion & r, int lane)
{
    const int g_lo = (r.x1 + 7) >> 3, g_hi = (r.x2 + 1) >> 3;
    const uint32_t rh = (uint32_t)(r.y2 - r.y1 + 1);
    int a_end = r.x2 + 1, b_start = r.x2 + 1;  // columns [x1, a_end) and [b_start, x2] go pixel by pixel
    if(g_hi > g_lo) {
        const uint32_t ng = (uint32_t)(g_hi - g_lo), n = ng * rh;
        const float inv = 1.0f / (float)ng;
        for(uint32_t idx = (uint32_t)lane; idx < n; idx += 64) {
            const uint32_t ly = small_div(idx, inv), lg = idx - ly * ng;
            const int row = r.y1 + (int)ly, g = g_lo + (int)lg;
            argb_list_group<SRC>(it, swap, tile, f, row, g, g * 8 + r.ox, row + r.oy);
        }
        a_end = g_lo * 8;
        b_start = g_hi * 8;
    }
    const uint32_t wa = (uint32_t)(a_end - r.x1), wb = (uint32_t)(r.x2 + 1 - b_start);
    const uint32_t rw = wa + wb, n = rw * rh;
    if(n == 0) return;
    const float inv = 1.0f / (float)rw;
    for(uint32_t idx = (uint32_t)lane; idx < n; idx += 64) {
        const uint32_t ly = small_div(idx, inv), lx = idx - ly * rw;
        const int col = lx < wa ? r.x1 + (int)lx : b_start + (int)(lx - wa), row = r.y1 + (int)ly;
        argb_list_px<SRC>(it, swap, tile, f, row, col, col + r.ox, row + r.oy);
    }
}

template <int SRC>
__device__ __forceinline__ void argb_list_batch(const ListArgs & a, uint8_t * tile, uint32_t f, int tx, int ty, uint32_t first,
                                                uint32_t n_full, uint32_t n_part, int lane)
{
    const uint4 * ch = a.chunks + first;
    const uint32_t n = n_full + n_part;
    for(uint32_t base = 0; base < n; base += 64) {
        const uint32_t idx = base + (uint32_t)lane;
        if(idx >= n) continue;
        const uint4 c = ch[idx];
        const ListItem & it = a.items[c.x];
        const int row = (int)(c.y & 63u), g = (int)((c.y >> 6) & 7u);
        const uint32_t pxmask = (c.y >> 9) & 0xFFu;
        const int ax = tx * LT + g * 8 - it.x1, ay = ty * LT + row - it.y1;
        if(idx < n_full) argb_list_group<SRC>(it, a.swap, tile, f, row, g, ax, ay);
        else {
#pragma unroll
            for(int j = 0; j < 8; j++)
                if((pxmask >> j) & 1u) argb_list_px<SRC>(it, a.swap, tile, f, row, g * 8 + j, ax + j, ay);
        }
    }
}

__global__ __launch_bounds__(256) void k_blend_list_argb(ListArgs a)
{
    __shared__ uint4 lds4[4][LT * LT * 3 / 16];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const uint32_t wt = __builtin_amdgcn_readfirstlane(xcd_remap(blockIdx.x, gridDim.x) * 4 + wave);
    if(wt >= a.n_sel * a.n_frames) return;
    const uint32_t f = wt / a.n_sel;
    const uint32_t sel = a.sel[wt - f * a.n_sel];
    const uint32_t t = sel & 0x7FFFFFFFu;
    const bool overwrite = (sel >> 31) != 0;
    const int ty = (int)(t / a.tiles_x), tx = (int)(t - (uint32_t)ty * a.tiles_x);
    const int W = a.W, H = a.H;
    uint8_t * dst = reinterpret_cast<uint8_t *>(a.dst) + (long long)f * a.dst_fs * 3;
    uint4 * tile4 = lds4[wave];
    uint8_t * tile = reinterpret_cast<uint8_t *>(tile4);
    const uint32_t k0 = a.bin_start[t], k1 = a.bin_start[t + 1];
    // a tile row is 192 bytes = 12 chunks of 16; the tile's part of the frame may be narrower (W % 16 == 0)
    const int tw = min(LT, W - tx * LT), th = min(LT, H - ty * LT);
    const int cpr = tw * 3 / 16;               // 16-byte chunks per row inside the frame
    const int n_chunks = cpr * th;
    if(!overwrite) {
        for(int i = lane; i < n_chunks; i += 64) {
            const int row = i / cpr, c = i - row * cpr;
            tile4[row * 12 + c] = *reinterpret_cast<const uint4 *>(dst + ((size_t)(ty * LT + row) * W + tx * LT) * 3 + c * 16);
        }
    }
    wave_lds_fence();

    uint4 st_next = k0 < k1 ? a.steps[k0] : make_uint4(0, 0, 0, 0);
    for(uint32_t k = k0; k < k1; k++) {
        const uint32_t s_x = __builtin_amdgcn_readfirstlane(st_next.x), s_y = __builtin_amdgcn_readfirstlane(st_next.y);
        const uint32_t s_z = __builtin_amdgcn_readfirstlane(st_next.z), s_w = __builtin_amdgcn_readfirstlane(st_next.w);
        if(k + 1 < k1) st_next = a.steps[k + 1];
        if(s_x & LIST_STEP_ITEM) {
            const ListItem & it = a.items[s_x & ~LIST_STEP_ITEM];
            ListRegion r;
            r.x1 = max(it.x1, tx * LT) - tx * LT;
            r.x2 = min(it.x2, tx * LT + LT - 1) - tx * LT;
            r.y1 = max(it.y1, ty * LT) - ty * LT;
            r.y2 = min(it.y2, ty * LT + LT - 1) - ty * LT;
            if(r.x2 < r.x1 || r.y2 < r.y1) continue;
            r.ox = tx * LT - it.x1;
            r.oy = ty * LT - it.y1;
            switch(it.fmt) {
                case SRC_FILL: argb_list_item<SRC_FILL>(it, a.swap, tile, f, r, lane); break;
                case SRC_RGB565: argb_list_item<SRC_RGB565>(it, a.swap, tile, f, r, lane); break;
                case SRC_ARGB8888: argb_list_item<SRC_ARGB8888>(it, a.swap, tile, f, r, lane); break;
                case SRC_RGBA8888: argb_list_item<SRC_RGBA8888>(it, a.swap, tile, f, r, lane); break;
                default: argb_list_item<SRC_ARGB8565>(it, a.swap, tile, f, r, lane); break;
            }
        }
        else {
            switch(s_x & 7u) {
                case SRC_FILL: argb_list_batch<SRC_FILL>(a, tile, f, tx, ty, s_y, s_z, s_w, lane); break;
                case SRC_RGB565: argb_list_batch<SRC_RGB565>(a, tile, f, tx, ty, s_y, s_z, s_w, lane); break;
                case SRC_ARGB8888: argb_list_batch<SRC_ARGB8888>(a, tile, f, tx, ty, s_y, s_z, s_w, lane); break;
                case SRC_RGBA8888: argb_list_batch<SRC_RGBA8888>(a, tile, f, tx, ty, s_y, s_z, s_w, lane); break;
                default: argb_list_batch<SRC_ARGB8565>(a, tile, f, tx, ty, s_y, s_z, s_w, lane); break;
            }
        }
        wave_lds_fence();
    }

    for(int i = lane; i < n_chunks; i += 64) {
        const int row = i / cpr, c = i - row * cpr;
        *reinterpret_cast<uint4 *>(dst + ((size_t)(ty * LT + row) * W + tx * LT) * 3 + c * 16) = tile4[row * 12 + c];
    }
}

}  // namespace lvgpu
