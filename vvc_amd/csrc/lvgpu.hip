// lvgpu.hip -- C ABI of liblvgpu.so (see include/lvgpu.h): argument checking, the control logic of
// lv_draw_sw_blend / lv_draw_sw_blend_basic (libraries/lvgl/src/draw/sw/lv_draw_sw_blend.c:95-204) resolved on the
// host into kernel arguments, and kernel selection.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/lvgpu.h"
#include "lvgpu_launch.hpp"
#include "lvgpu_list.hpp"

using namespace lvgpu;

// Scratch of lvgpu_blend_batch_noalloc: per frame, the scan index of the first non-black pixel of the area (what
// prepare_first_nb otherwise keeps in a stream-ordered allocation).  Every GPU holds its own copy of the array until
// the process ends.  In host code the name below is only a handle for hipGetSymbolAddress: the address kernels and
// hipMemsetAsync get is the per-device pointer kept in g_scratch_ptr.
static __device__ __align__(256) uint32_t scratch[LVGPU_NOALLOC_MAX_FRAMES];

namespace {

thread_local std::string g_err;
thread_local const char * g_last_kernel = "";
std::atomic<uint64_t> g_alloc_calls{0}, g_first_nb_launches{0};  // lvgpu_debug_counters

int fail(int code, const char * fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

// A failed runtime call also stays behind as the thread's "last error", which the check after the next kernel launch
// would report as that launch's: it is taken off here, so that a host can carry on after, say, a refused allocation.
#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if(e_ != hipSuccess) {                                                                     \
            (void)hipGetLastError();                                                               \
            return fail(LVGPU_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_));                    \
        }                                                                                          \
    } while(0)
}  // namespace

namespace lvgpu {
int host_fail(int code, const char * text) { return fail(code, "%s", text); }
}  // namespace lvgpu

namespace {

inline int aw(const lvgpu_area_t & a) { return a.x2 - a.x1 + 1; }
inline int ah(const lvgpu_area_t & a) { return a.y2 - a.y1 + 1; }
inline bool intersect(lvgpu_area_t & r, const lvgpu_area_t & a, const lvgpu_area_t & b)  // lv_area.c:129-144
{
    r.x1 = a.x1 > b.x1 ? a.x1 : b.x1;
    r.y1 = a.y1 > b.y1 ? a.y1 : b.y1;
    r.x2 = a.x2 < b.x2 ? a.x2 : b.x2;
    r.y2 = a.y2 < b.y2 ? a.y2 : b.y2;
    return !(r.x1 > r.x2 || r.y1 > r.y2);
}
inline uint32_t bswap16(uint32_t v) { return ((v << 8) | (v >> 8)) & 0xFFFFu; }
inline uint32_t host_mix(uint32_t fg, uint32_t bg, uint32_t mix)  // native values; lv_color.h:447-453
{
    uint32_t m5 = (mix + 4) >> 3;
    uint32_t b = (bg | (bg << 16)) & 0x07E0F81Fu, f = (fg | (fg << 16)) & 0x07E0F81Fu;
    uint32_t r = ((((f - b) * m5) >> 5) + b) & 0x07E0F81Fu;
    return ((r >> 16) | r) & 0xFFFFu;
}
inline bool aligned(const void * p, size_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; }

int src_bytes_per_px(int fmt)
{
    switch(fmt) {
        case LVGPU_SRC_RGB565: return 2;
        case LVGPU_SRC_ARGB8888:
        case LVGPU_SRC_RGBA8888: return 4;
        case LVGPU_SRC_ARGB8565: return 3;
        default: return 0;
    }
}

// Device address of `scratch` on the current device, looked up once per device with that device current.
constexpr int MAX_DEVICES = 64;
uint32_t * g_scratch_ptr[MAX_DEVICES];
std::mutex g_scratch_mutex;
int scratch_on_current_device(uint32_t ** out)
{
    int dev = -1;
    HIP_TRY(hipGetDevice(&dev));
    if(dev < 0 || dev >= MAX_DEVICES) return fail(LVGPU_ERR_ARG, "device %d: only devices 0..%d are supported", dev, MAX_DEVICES - 1);
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if(g_scratch_ptr[dev] == nullptr) {
        uint32_t * p = nullptr;
        HIP_TRY(hipGetSymbolAddress(reinterpret_cast<void **>(&p), HIP_SYMBOL(scratch)));
        if(p == nullptr) return fail(LVGPU_ERR_HIP, "hipGetSymbolAddress(scratch) returned NULL on device %d", dev);
        g_scratch_ptr[dev] = p;
    }
    *out = g_scratch_ptr[dev];
    return LVGPU_OK;
}

enum Resolve { R_NOOP = 0, R_WORK = 1 };

// lv_draw_sw_blend (:95-106) + lv_draw_sw_blend_basic (:108-204) up to the fill_normal/map_normal dispatch.
// Returns R_NOOP / R_WORK or a negative error; on R_WORK `a` describes the clipped area.
int resolve_blend(const lvgpu_blend_batch_t * d, BlendArgs & a, lvgpu_area_t & area, unsigned allowed_flags = 0)
{
    allowed_flags |= LVGPU_BLEND_FLAG_NO_AA;
    memset(&a, 0, sizeof(a));
    if(d == nullptr) return fail(LVGPU_ERR_ARG, "descriptor is NULL");
    if(d->n_frames < 0) return fail(LVGPU_ERR_ARG, "n_frames < 0");
    if(d->n_frames == 0) return R_NOOP;
    if(d->dst == nullptr) return fail(LVGPU_ERR_ARG, "dst is NULL");
    if(d->src_format < LVGPU_SRC_FILL || d->src_format > LVGPU_SRC_ARGB8565)
        return fail(LVGPU_ERR_ARG, "bad src_format %d", d->src_format);
    if(d->src_format != LVGPU_SRC_FILL && d->src == nullptr) return fail(LVGPU_ERR_ARG, "src is NULL");
    if(aw(d->buf_area) <= 0 || ah(d->buf_area) <= 0) return fail(LVGPU_ERR_ARG, "empty buf_area");
    const bool own_alpha = d->src_format >= LVGPU_SRC_ARGB8888;
    if(own_alpha && d->mask) return fail(LVGPU_ERR_UNSUPPORTED, "a mask on top of a source with its own alpha");
    if(d->blend_mode > LVGPU_BLEND_REPLACE) return fail(LVGPU_ERR_ARG, "bad blend_mode %d", d->blend_mode);
    if(d->flags & ~allowed_flags) return fail(LVGPU_ERR_ARG, "flags 0x%02x: this entry point takes 0x%02x only", d->flags, allowed_flags);

    if(d->opa <= 2) return R_NOOP;                                            // :98  LV_OPA_MIN
    if(!intersect(area, d->blend_area, d->clip_area)) return R_NOOP;          // :101
    if((int64_t)aw(area) * ah(area) > 0x7FFFFFFFll) return fail(LVGPU_ERR_ARG, "area larger than 2^31 pixels");
    // The reference trusts clip_area to lie inside buf_area; a GPU write outside the buffer is not acceptable.
    if(area.x1 < d->buf_area.x1 || area.y1 < d->buf_area.y1 || area.x2 > d->buf_area.x2 || area.y2 > d->buf_area.y2)
        return fail(LVGPU_ERR_ARG, "clip_area reaches outside buf_area");

    const uint8_t * mask = d->mask;
    uint8_t mask_res = d->mask_res;
    if(own_alpha) mask_res = LVGPU_MASK_RES_CHANGED;                          // lv_draw_sw_img.c:133-136
    if(!own_alpha) {
        if(mask && mask_res == LVGPU_MASK_RES_TRANSP) return R_NOOP;          // :113
        else if(mask_res == LVGPU_MASK_RES_FULL_COVER) mask = nullptr;        // :114
    }
    const bool masked = own_alpha || mask != nullptr;
    if(mask) {
        if(area.x1 < d->mask_area.x1 || area.y1 < d->mask_area.y1 || area.x2 > d->mask_area.x2 ||
           area.y2 > d->mask_area.y2)
            return fail(LVGPU_ERR_ARG, "mask_area does not cover the clipped blend area");
    }

    a.dst_stride = aw(d->buf_area);                                           // :117
    a.dst_fs = d->dst_frame_stride;
    a.dst = reinterpret_cast<uint16_t *>(d->dst) +
            (size_t)a.dst_stride * (area.y1 - d->buf_area.y1) + (area.x1 - d->buf_area.x1);   // :126
    a.w = aw(area);
    a.h = ah(area);
    a.n_frames = d->n_frames;
    a.swap = d->color_16_swap ? 1 : 0;
    a.opa = d->opa;
    if(d->src_format != LVGPU_SRC_FILL) {
        a.src_stride = aw(d->blend_area);                                     // :140
        a.src_fs = d->src_frame_stride;
        size_t ofs = (size_t)a.src_stride * (area.y1 - d->blend_area.y1) + (area.x1 - d->blend_area.x1);   // :141
        a.src = reinterpret_cast<const uint8_t *>(d->src) + ofs * src_bytes_per_px(d->src_format);
    }
    if(mask) {
        a.mask_stride = aw(d->mask_area);                                     // :158
        a.mask_fs = d->mask_frame_stride;
        a.mask = mask + (size_t)a.mask_stride * (area.y1 - d->mask_area.y1) + (area.x1 - d->mask_area.x1);   // :159
    }
    a.color = a.swap ? bswap16(d->color) : d->color;
    a.raw_opa = d->opa;
    a.raw_mode = d->blend_mode;
    a.has_mask = masked ? 1 : 0;
    a.no_aa = (masked && (d->flags & LVGPU_BLEND_FLAG_NO_AA)) ? 1 : 0;                // :150-156
    if(d->dst_argb8565) {  // :177-184: fill_argb / map_argb whatever the blend mode
        a.dst = reinterpret_cast<uint16_t *>(reinterpret_cast<uint8_t *>(d->dst) +
                ((size_t)a.dst_stride * (area.y1 - d->buf_area.y1) + (area.x1 - d->buf_area.x1)) * 3);   // :129-133
        return R_WORK;
    }

    const bool fill = d->src_format == LVGPU_SRC_FILL;
    if(d->blend_mode == LVGPU_BLEND_NORMAL) {                                 // :186-193
        a.blend_fn = BF_NORMAL;
        if(!masked) {
            if(d->opa >= 253) {                                               // fill :254-259, map :655-660
                a.opa_mode = OM_CONST;
                a.opa = 255;
            }
            else if(fill) {                                                   // :261-287
                a.opa_mode = OM_FILL_PREMULT;
                const uint32_t opa_q = ((((uint32_t)d->opa + 4) >> 3) << 3) & 0xFFu;   // uint8 wrap: 252 -> 0
                const uint32_t c = a.color;
                a.pf.pm_r = (c >> 11) * opa_q;
                a.pf.pm_g = ((c >> 5) & 0x3Fu) * opa_q;
                a.pf.pm_b = (c & 0x1Fu) * opa_q;
                a.pf.opa_inv = 255u - opa_q;
                a.pf.first_res = host_mix(c, 0, d->opa);
            }
            else a.opa_mode = OM_CONST;                                       // :662-670
        }
        else if(fill) a.opa_mode = d->opa >= 253 ? OM_MASK : OM_FILL_MASK_OPA;   // :295 / :346
        else a.opa_mode = d->opa > 253 ? OM_MASK : OM_MAP_MASK_OPA;           // :675 (strict) / :720
    }
    else if(d->blend_mode <= LVGPU_BLEND_MULTIPLY) {                          // :194-203, fill_blended / map_blended
        a.blend_fn = d->blend_mode;
        a.opa_mode = masked ? OM_MAP_MASK_OPA : OM_CONST;                     // :583/:590, :929/:934
    }
    else return R_NOOP;  // "unsupported blend mode": the reference logs a warning and draws nothing (:556-558)
    return R_WORK;
}

// min index of the first non-black pixel per frame, only needed when the two formulas of the translucent fill
// disagree on black (otherwise every black pixel gets the same value either way).
// provided: device memory for n_frames indices that the caller owns (nothing is allocated then and *owned stays
// nullptr); otherwise the indices live in a stream-ordered allocation that the caller frees (hipFreeAsync) after the
// blend kernel has been launched.
int prepare_first_nb(BlendArgs & a, hipStream_t st, uint32_t ** owned, uint32_t * provided = nullptr)
{
    *owned = nullptr;
    if(a.opa_mode != OM_FILL_PREMULT) return LVGPU_OK;
    const uint32_t pm_black = ((((a.pf.pm_r * 0x8081u) >> 23) & 0x1Fu) << 11) | ((((a.pf.pm_g * 0x8081u) >> 23) & 0x3Fu) << 5) |
                              (((a.pf.pm_b * 0x8081u) >> 23) & 0x1Fu);
    if(pm_black == a.pf.first_res) return LVGPU_OK;
    uint32_t * buf = provided;
    if(buf == nullptr) {
        g_alloc_calls++;
        HIP_TRY(hipMallocAsync(reinterpret_cast<void **>(&buf), (size_t)a.n_frames * sizeof(uint32_t), st));
        *owned = buf;
    }
    hipError_t e = hipMemsetAsync(buf, 0xFF, (size_t)a.n_frames * sizeof(uint32_t), st);
    if(e == hipSuccess) {
        // one block per frame when there are frames enough to fill the chip, otherwise several segments per frame
        // (an all-black 4K canvas is 16 MB for its blocks to walk); a segment is a whole number of 1024-pixel pieces
        const uint32_t area = (uint32_t)a.w * (uint32_t)a.h;
        const uint32_t pieces = (area + 1023u) / 1024u;
        const uint32_t want = std::max<uint32_t>(1u, std::min<uint32_t>((2048u + a.n_frames - 1) / a.n_frames, (pieces + 3u) / 4u));
        const uint32_t seg_px = ((pieces + want - 1) / want) * 1024u;
        dim3 grid((area + seg_px - 1) / seg_px, std::min<uint32_t>(a.n_frames, 65535));
        hipLaunchKernelGGL(k_first_nonblack, grid, dim3(256), 0, st, a.dst, a.dst_fs, a.dst_stride, a.w, a.h, a.n_frames, seg_px, buf);
        e = hipGetLastError();
        g_first_nb_launches++;
    }
    if(e != hipSuccess) {
        if(*owned) (void)hipFreeAsync(*owned, st);
        *owned = nullptr;
        return fail(LVGPU_ERR_HIP, "first-non-black pass: %s", hipGetErrorString(e));
    }
    a.first_nb = buf;
    return LVGPU_OK;
}

// How many workgroups of the full-frame blend kernels share a CU decides how many independent address streams the
// memory system sees at once.  With the whole device streaming, fewer residents reach a higher HBM rate for the
// plain-mix case (few instructions per byte): the linear kernel 70.3 -> 73.7 % of peak, the fused tile kernel
// 66.4 -> 70.0 %, the rotation alone 70.4 -> 74.5 %.  Residency is capped at launch by asking for dynamic LDS the
// kernels never touch; the values below sit on measured plateaus of the rate against the requested size (a masked
// fill, with less to load per lane, likes more residents than an image; the RGB565 + A8 source and 4K canvases fall
// off right above the tile kernel's value).  On a stream confined to some of the CUs (a CU mask) the same cap costs
// 10-30 %, and the instruction-heavier blend functions lose with it everywhere, so it is applied only to the
// launches named at its uses, on streams that own the whole device.
}  // namespace
namespace lvgpu {
uint32_t residency_pad(hipStream_t st, uint32_t pad_bytes)
{
    // LVGPU_NO_RESIDENCY_CAP in the environment switches the cap off: tools/check_residency.py uses it to verify, on
    // the machine at hand, that every capped launch is still at least as fast as the uncapped one.
    static const bool disabled = getenv("LVGPU_NO_RESIDENCY_CAP") != nullptr;
    if(disabled) return 0;
    static thread_local int dev_cached = -1, dev_cus = 0;
    int dev = -1;
    if(hipGetDevice(&dev) != hipSuccess) return 0;
    if(dev != dev_cached) {
        if(hipDeviceGetAttribute(&dev_cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
        dev_cached = dev;
    }
    uint32_t mask[32] = {0};
    if(hipExtStreamGetCUMask(st, 32, mask) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    int cus = 0;
    for(uint32_t m : mask) cus += __builtin_popcount(m);
    return cus >= dev_cus ? pad_bytes : 0;
}

}  // namespace lvgpu
namespace {
template <int SRC>
struct LaunchLinear {
    static int run(const BlendArgs & a, hipStream_t st)
    {
        const uint32_t gpf = (uint32_t)(((size_t)a.w * a.h) / 8);
        const uint32_t cpf = (gpf + LIN_CHUNK - 1) / LIN_CHUNK;
        const uint64_t blocks = (uint64_t)cpf * a.n_frames;
        if(blocks > 0x7FFFFFFFull) return fail(LVGPU_ERR_ARG, "batch too large for one launch");
        const int fast = fast_path<SRC>(a);
        uint32_t pad = 0;
        if(fast_is_mix(fast)) pad = residency_pad(st, SRC == SRC_FILL ? LIN_PAD_LDS_FILL : LIN_PAD_LDS);
        // the translucent unmasked fill has twice the instructions per byte: six workgroups per CU instead of eight gain
        // 1.5 points (67.5 -> 69.1 % of the HBM peak), five and fewer lose (62 %, 53 %, 39 %)
        else if(fast == FAST_PREMULT) pad = residency_pad(st, LIN_PAD_LDS_PREMULT);
        if(const char * e = getenv("LVGPU_LIN_PAD")) pad = residency_pad(st, (uint32_t)atoi(e));   // for measuring
        // FAST_OPAQUE (opaque fill / map: write-once stores that bypass the L2) is fastest uncapped: 88 % / 77 % of peak
        hipLaunchKernelGGL(k_blend_linear<SRC>, dim3((uint32_t)blocks), dim3(256), pad, st, a, gpf, cpf);
        HIP_TRY(hipGetLastError());
        return LVGPU_OK;
    }
};
template <int SRC>
struct LaunchArgbDst {
    static int run(const BlendArgs & a, hipStream_t st)
    {
        const uint32_t area = (uint32_t)a.w * (uint32_t)a.h;
        dim3 grid((area + 255) / 256, std::min<uint32_t>(a.n_frames, 65535));
        hipLaunchKernelGGL(k_blend_argb_dst<SRC>, grid, dim3(256), 0, st, a);
        HIP_TRY(hipGetLastError());
        return LVGPU_OK;
    }
};
template <int SRC>
struct LaunchArgbLinear {
    static int run(const BlendArgs & a, hipStream_t st)
    {
        const uint32_t gpf = (uint32_t)(((size_t)a.w * a.h) / 16);
        const uint32_t cpf = (gpf + ARGB_BLOCK_GROUPS - 1) / ARGB_BLOCK_GROUPS;
        const uint64_t blocks = (uint64_t)cpf * a.n_frames;
        if(blocks > 0x7FFFFFFFull) return fail(LVGPU_ERR_ARG, "batch too large for one launch");
        // Five workgroups per CU instead of eight where the destination is read and written (masked fill 67.9 -> 69.6 %
        // of the HBM peak, masked map at opacity 200 67.5 -> 68.8 %; six: 69.1 %, four: 67.9 %); a launch that only
        // writes (opaque unmasked fill or map) loses five points with it and has none.
        const bool blended = SRC != SRC_FILL && a.raw_mode >= 1 && a.raw_mode <= 3;
        const bool need_dst = a.has_mask || a.raw_opa < 253 || blended;   // as in the kernel
        uint32_t pad = need_dst ? residency_pad(st, ARGB_PAD_LDS) : 0;
        if(const char * e = getenv("LVGPU_ARGB_PAD")) pad = residency_pad(st, (uint32_t)atoi(e));   // for measuring
        hipLaunchKernelGGL(k_blend_argb_linear<SRC>, dim3((uint32_t)blocks), dim3(256), pad, st, a, gpf, cpf);
        HIP_TRY(hipGetLastError());
        return LVGPU_OK;
    }
};
// Panel rows of a 90/270-degree output are 2*H bytes long: with H % 32 != 0 they start off 64-byte boundaries and
// the tile kernel's 128-byte output runs degrade; the column kernel writes runs of up to 2 KB.  With H % 64 == 32 the
// rows start on 64-byte boundaries, every other of the tile kernel's runs straddles two lines and its last row of tiles
// is half empty: the column kernel is ahead there too from 160 rows on (640x480 fused 66.3 -> 67.9 % of the HBM peak,
// rotation alone 67.9 -> 71.2 %; 1280x800 65.1 -> 68.5 %, 65.9 -> 71.6 %; 2560x1440 63.8 -> 66.9 %, 63.9 -> 71.1 %;
// 128x160 67.7 -> 70.5 %, 69.0 -> 72.6 %), not on 96 rows fused (65.9 against 55.2 %).
inline bool wants_col(const BlendArgs & a)
{
    if(const char * e = getenv("LVGPU_QT_KERNEL")) {   // a switch for measuring, see tile_ok
        if(e[0] == 'c') return true;
        if(e[0] == 't') return false;
    }
    return (a.h % 32) != 0 || ((a.h % 64) != 0 && a.h > 128);
}

// LVGPU_NO_BAND in the environment keeps masked fused launches on the tile kernel: tools/bench_fused_src.py uses it to
// compare the two on the machine at hand.
inline bool band_disabled()
{
    static const bool off = getenv("LVGPU_NO_BAND") != nullptr;
    return off;
}

// Can the clipped area of every frame be walked as one contiguous, 16-byte aligned run of 8-pixel groups?
bool linear_ok(const BlendArgs & a, int fmt)
{
    if(((size_t)a.w * a.h) % 8) return false;
    if(a.h > 1 && a.dst_stride != a.w) return false;
    if(!aligned(a.dst, 16) || (a.dst_fs % 8)) return false;
    if(fmt != SRC_FILL && fmt != SRC_COPY) {
        if(a.h > 1 && a.src_stride != a.w) return false;
        const size_t bpp = src_bytes_per_px(fmt);
        const size_t al = fmt == SRC_ARGB8565 ? 8 : 16;
        if(!aligned(a.src, al) || ((a.src_fs * bpp) % al)) return false;
    }
    if(a.mask) {
        if(a.h > 1 && a.mask_stride != a.w) return false;
        if(!aligned(a.mask, 8) || (a.mask_fs % 8)) return false;
    }
    return true;
}
// ... and additionally as whole W x H frames for the tile kernel (row starts stay 16-byte aligned).
bool tile_ok(const BlendArgs & a, int fmt)
{
    // Heights off the 32-pixel grid: which of the two kernels that write long panel-row runs is ahead was measured.
    // With the blend fused in, k_tile_any up to 256 rows (one run per panel row: 320x200 68 against 63 % of peak), the
    // k_tile_col above (480x328 67 against 53 %, 1080p 64 against 56 %).  The rotation alone stays with k_tile_any up to
    // 384 rows (level with k_tile_col there) and from 1281 to 1536 rows; from 385 to 1280 rows, where one k_tile_col
    // workgroup takes a column of the frame whole, and on 4K canvases k_tile_col is ahead (1024x600 71 against 64 %,
    // 1280x720 70 against 66 %, 1080p 69 against 67 %, 3840x2160 67 against 65 %).
    if((a.h % 32) != 0) {
        // LVGPU_QT_KERNEL=any|col|tile in the environment (read at every call) overrides the choice: a switch for measuring
        if(const char * e = getenv("LVGPU_QT_KERNEL")) {
            if(e[0] == 'a') return false;
        }
        else {
            if(a.h <= 256) return false;
            if(fmt == SRC_COPY && (a.h <= 384 || (a.h > 1280 && a.h <= 1536))) return false;
        }
    }
    return linear_ok(a, fmt) && (a.w % 8) == 0 && (a.h % 8) == 0;
}

// The same question for a 3-byte destination: groups of 16 pixels are 48 bytes = three 16-byte words.
bool argb_linear_ok(const BlendArgs & a, int fmt)
{
    if(((size_t)a.w * a.h) % 16) return false;
    if(a.h > 1 && a.dst_stride != a.w) return false;
    if(!aligned(a.dst, 16) || ((a.dst_fs * 3) % 16)) return false;
    if(fmt != SRC_FILL) {
        if(a.h > 1 && a.src_stride != a.w) return false;
        const size_t bpp = src_bytes_per_px(fmt);
        const size_t al = fmt == SRC_ARGB8565 ? 8 : 16;
        if(!aligned(a.src, al) || ((a.src_fs * bpp) % al)) return false;
    }
    if(a.mask) {
        if(a.h > 1 && a.mask_stride != a.w) return false;
        if(!aligned(a.mask, 16) || (a.mask_fs % 16)) return false;
    }
    return true;
}

int launch_blend(BlendArgs & a, int fmt, hipStream_t st, bool argb_dst = false, uint32_t * first_nb_buf = nullptr)
{
    if(argb_dst) {
        if(argb_linear_ok(a, fmt)) {
            g_last_kernel = "k_blend_argb_linear";
            return dispatch_src<LaunchArgbLinear>(fmt, a, st);
        }
        // LVGPU_ARGB_KERNEL=px in the environment: the one-pixel-per-lane kernel instead (for measuring and for tests
        // that compare the two; read at every call)
        const char * e = getenv("LVGPU_ARGB_KERNEL");
        if(e && !strcmp(e, "px")) {
            g_last_kernel = "k_blend_argb_dst";
            return dispatch_src<LaunchArgbDst>(fmt, a, st);
        }
        g_last_kernel = "k_blend_argb_rows";
        return launch_argb_rows(fmt, a, st);
    }
    uint32_t * owned = nullptr;
    int rc = prepare_first_nb(a, st, &owned, first_nb_buf);
    if(rc != LVGPU_OK) return rc;
    if(linear_ok(a, fmt)) {
        g_last_kernel = "k_blend_linear";
        rc = dispatch_src<LaunchLinear>(fmt, a, st);
    }
    else {
        g_last_kernel = "k_blend_rows";
        rc = launch_rows(fmt, a, st);
    }
    if(owned) {  // in stream order: after the kernels that read it
        const hipError_t e_free = hipFreeAsync(owned, st);
        if(rc == LVGPU_OK && e_free != hipSuccess) return fail(LVGPU_ERR_HIP, "hipFreeAsync: %s", hipGetErrorString(e_free));
    }
    return rc;
}

int launch_rotate(const uint16_t * in, long long in_fs, uint16_t * out, long long out_fs, int W, int H, int rot,
                  int swap_out, int n_frames, hipStream_t st)
{
    BlendArgs a;
    memset(&a, 0, sizeof(a));
    a.dst = const_cast<uint16_t *>(in);
    a.dst_fs = in_fs;
    a.dst_stride = W;
    a.w = W;
    a.h = H;
    a.n_frames = n_frames;
    a.out = out;
    a.out_fs = out_fs;
    a.rot = rot;
    a.swap_out = swap_out ? 1 : 0;
    const bool out_ok = aligned(out, 16) && (out_fs % 8) == 0;
    const bool transposing = rot == LVGPU_ROT_90 || rot == LVGPU_ROT_270;
    if(out_ok && transposing && tile_ok(a, SRC_COPY)) {
        if(wants_col(a)) {
            g_last_kernel = "k_tile_col";
            return launch_tile_col(SRC_COPY, a, st);
        }
        g_last_kernel = "k_tile";
        return launch_tile(SRC_COPY, a, st);
    }
    if(out_ok && !transposing && linear_ok(a, SRC_COPY)) {
        g_last_kernel = "k_blend_linear";
        return LaunchLinear<SRC_COPY>::run(a, st);
    }
    if(transposing) {  // any size, any alignment
        g_last_kernel = "k_tile_any";
        return launch_tile_any(SRC_COPY, a, st);
    }
    g_last_kernel = "k_flush_linear_any";
    const uint32_t area = (uint32_t)W * (uint32_t)H;
    dim3 grid(((area + 7) / 8 + 255) / 256, std::min<uint32_t>(n_frames, 65535));
    hipLaunchKernelGGL(k_flush_linear_any, grid, dim3(256), 0, st, in, in_fs, out, out_fs, area, rot, swap_out ? 1 : 0, n_frames);
    HIP_TRY(hipGetLastError());
    return LVGPU_OK;
}

// grid of the 8-pixels-per-lane conversion kernels: one group per lane
inline dim3 vector_grid(size_t n8) { return dim3((uint32_t)std::min<size_t>(std::max<size_t>((n8 + 255) / 256, 1), 0x7FFFFFFFull)); }
typedef void (*convert_kernel_t)(const void *, uint4 *, uint2 *, size_t, int, int, uint32_t, int, PremultFill);
inline convert_kernel_t vector_convert_kernel(int fmt)
{
    switch(fmt) {
        case LVGPU_SRC_RGB565: return k_image_convert_v<SRC_RGB565>;
        case LVGPU_SRC_ARGB8888: return k_image_convert_v<SRC_ARGB8888>;
        case LVGPU_SRC_RGBA8888: return k_image_convert_v<SRC_RGBA8888>;
        default: return k_image_convert_v<SRC_ARGB8565>;
    }
}
inline dim3 stream_grid(size_t n)
{
    size_t b = (n + 255) / 256;
    return dim3((uint32_t)std::min<size_t>(std::max<size_t>(b, 1), 256 * 16));
}

// ---------------------------------------------------------------------------------------------------------------
// display lists
// ---------------------------------------------------------------------------------------------------------------
bool needs_scan_order(const BlendArgs & a)
{
    if(a.opa_mode != OM_FILL_PREMULT) return false;
    const uint32_t pm_black = ((((a.pf.pm_r * 0x8081u) >> 23) & 0x1Fu) << 11) | ((((a.pf.pm_g * 0x8081u) >> 23) & 0x3Fu) << 5) |
                              (((a.pf.pm_b * 0x8081u) >> 23) & 0x1Fu);
    return pm_black != a.pf.first_res;
}

// Items the tile pass does not take: a translucent fill that needs its scan-order pre-pass, and blends with
// anti-aliasing off (the tile kernels are compiled without that switch; LVGL's own callback never sets it, it rounds
// the caller's masks itself).  They run through the single-descriptor path in their place in the order.
bool needs_own_pass(const BlendArgs & a) { return needs_scan_order(a) || a.no_aa != 0; }

struct ListBuilder {
    int W, H;
    uint32_t tiles_x, tiles_y;
    std::vector<ListItem> items;
    std::vector<std::vector<uint32_t>> bins;
    bool argb;  // ARGB8565 destination
    ListBuilder(int w, int h, bool argb_dst)
        : W(w), H(h), tiles_x((w + LT - 1) / LT), tiles_y((h + LT - 1) / LT), bins((size_t)tiles_x * tiles_y), argb(argb_dst)
    {
    }
    void add(const BlendArgs & a, int fmt, const lvgpu_area_t & area, const lvgpu_area_t & buf)
    {
        ListItem it;
        memset(&it, 0, sizeof(it));
        it.src = a.src;
        it.mask = a.mask;
        it.src_fs = a.src_fs;
        it.mask_fs = a.mask_fs;
        it.src_stride = a.src_stride;
        it.mask_stride = a.mask_stride;
        it.x1 = area.x1 - buf.x1;
        it.y1 = area.y1 - buf.y1;
        it.x2 = area.x2 - buf.x1;
        it.y2 = area.y2 - buf.y1;
        it.color = a.color;
        it.opa = a.opa;
        it.opa_mode = a.opa_mode;
        it.blend_fn = a.blend_fn;
        it.fmt = fmt;
        it.swap = a.swap;
        it.pf = a.pf;
        it.raw_mode = a.raw_mode;
        it.has_mask = a.has_mask;
        if(argb) {
            it.opa = (uint32_t)a.raw_opa;
            // fill_argb's opaque masked loop reads row y of the area at mask[y*w + x] (lv_draw_sw_blend.c:494-505)
            if(fmt == SRC_FILL && a.mask && a.raw_opa >= 253) it.mask_stride = a.w;
        }
        const uint32_t id = (uint32_t)items.size();
        items.push_back(it);
        for(int ty = it.y1 / LT; ty <= it.y2 / LT; ty++)
            for(int tx = it.x1 / LT; tx <= it.x2 / LT; tx++) bins[(size_t)ty * tiles_x + tx].push_back(id);
    }
    // does item id replace every pixel of tile t without looking at it?
    bool overwrites(uint32_t t, uint32_t id) const
    {
        const ListItem & it = items[id];
        if(argb) {  // colour and alpha are stored whatever the pixel held: fill_argb :465-477, map_argb :768-790
            const bool blended = it.fmt != SRC_FILL && it.raw_mode >= 1 && it.raw_mode <= 3;
            if(it.has_mask || it.opa < 253u || blended) return false;
        }
        else {
            if(!(it.opa_mode == OM_CONST && it.opa == 255u && it.blend_fn == BF_NORMAL)) return false;
            if(!(it.fmt == SRC_FILL || it.fmt == SRC_RGB565)) return false;
        }
        const int tx = (int)(t % tiles_x), ty = (int)(t / tiles_x);
        const int x1 = tx * LT, y1 = ty * LT, x2 = std::min(x1 + LT, W) - 1, y2 = std::min(y1 + LT, H) - 1;
        return it.x1 <= x1 && it.y1 <= y1 && it.x2 >= x2 && it.y2 >= y2;
    }
    // Items of tile t that can still be seen: everything drawn before the last item that overwrites the whole tile
    // is dead in this tile.  Returns the position in bins[t] to start from; *ow tells whether that item overwrites.
    size_t first_live(uint32_t t, bool * ow) const
    {
        const std::vector<uint32_t> & b = bins[t];
        for(size_t i = b.size(); i-- > 0;) {
            if(overwrites(t, b[i])) {
                *ow = true;
                return i;
            }
        }
        *ow = false;
        return 0;
    }
    void clear()
    {
        items.clear();
        for(auto & b : bins) b.clear();
    }
};

// Pinned host memory through which a list's tables reach the device, one per calling thread, kept for the life of
// the thread.  The copy is asynchronous; the event tells the next call on this thread when the bytes have left.
struct ListStaging {
    uint8_t * host = nullptr;
    size_t cap = 0;
    hipEvent_t ev = nullptr;
    bool pending = false;
    int device = -1;  // the event belongs to the device that was current when it was created
    ~ListStaging()    // at thread exit: wait for a copy that may still read the buffer, then give everything back
    {
        if(pending && ev) (void)hipEventSynchronize(ev);
        if(ev) (void)hipEventDestroy(ev);
        if(host) (void)hipHostFree(host);
    }
    int acquire(size_t bytes, uint8_t ** out)
    {
        if(pending) {
            HIP_TRY(hipEventSynchronize(ev));
            pending = false;
        }
        int cur = -1;
        HIP_TRY(hipGetDevice(&cur));
        if(ev != nullptr && cur != device) {
            HIP_TRY(hipEventDestroy(ev));
            ev = nullptr;
        }
        if(ev == nullptr) {
            HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            device = cur;
        }
        if(bytes > cap) {
            if(host) HIP_TRY(hipHostFree(host));
            host = nullptr;
            cap = 0;
            const size_t want = std::max<size_t>(bytes * 2, 1 << 20);
            g_alloc_calls++;
            HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&host), want, hipHostMallocPortable));
            cap = want;
        }
        *out = host;
        return LVGPU_OK;
    }
    int release(hipStream_t st)
    {
        HIP_TRY(hipEventRecord(ev, st));
        pending = true;
        return LVGPU_OK;
    }
};
thread_local ListStaging g_stage;

// An item's part of a tile with fewer pixels than this is cut into chunks and batched with its neighbours in the
// list; larger parts are walked by the whole wave on their own.
constexpr int LIST_BIG_REGION = 512;

uint32_t item_kind(const ListItem & it, bool argb)
{
    if(argb) return (uint32_t)it.fmt | (it.has_mask ? 128u : 0u);  // every lane reads its own item's parameters
    const bool own_alpha = it.fmt == SRC_ARGB8888 || it.fmt == SRC_RGBA8888 || it.fmt == SRC_ARGB8565;
    const bool blended = it.blend_fn != BF_NORMAL;
    int om = blended ? 7 : it.opa_mode;
    if(!blended && it.opa_mode == OM_CONST && it.opa == 255u && !own_alpha) om = OM_OPAQUE;
    return list_kind(it.fmt, om, blended) | (it.mask ? 128u : 0u);
}

// The per-tile step lists of the accumulated items (see ListStep / ListChunk in lvgpu_list.hpp).
struct ListPlan {
    std::vector<uint32_t> bin_start, sel;
    std::vector<uint4> steps;
    std::vector<uint4> chunks;
    std::vector<uint4> full, part;  // the batch being collected (kept here so that their capacity survives)
    lvgpu_list_plan_t stats;
};

inline int popcount8(uint32_t v) { return __builtin_popcount(v & 0xFFu); }

// all_tiles: every tile is visited (the panel image is written from the tiles), otherwise only tiles with work.
void build_plan(const ListBuilder & lb, bool all_tiles, ListPlan & p)
{
    const uint32_t n_tiles = lb.tiles_x * lb.tiles_y;
    p.bin_start.assign(n_tiles + 1, 0);
    p.sel.clear();
    p.steps.clear();
    p.chunks.clear();
    memset(&p.stats, 0, sizeof(p.stats));
    p.stats.tiles = (int32_t)n_tiles;
    p.stats.items_listed = (int32_t)lb.items.size();
    std::vector<uint4> &full = p.full, &part = p.part;
    full.clear();
    part.clear();
    for(uint32_t t = 0; t < n_tiles; t++) {
        p.bin_start[t] = (uint32_t)p.steps.size();
        bool ow = false;
        const size_t first = lb.first_live(t, &ow);
        const int tx0 = (int)(t % lb.tiles_x) * LT, ty0 = (int)(t / lb.tiles_x) * LT;
        uint32_t kind = 0;
        uint64_t occ[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // per group column: rows of the tile the current batch touches
        auto flush = [&]() {
            if(full.empty() && part.empty()) return;
            p.steps.push_back(make_uint4(kind & 0x7Fu, (uint32_t)p.chunks.size(), (uint32_t)full.size(), (uint32_t)part.size()));
            p.chunks.insert(p.chunks.end(), full.begin(), full.end());
            p.chunks.insert(p.chunks.end(), part.begin(), part.end());
            p.stats.batch_steps++;
            p.stats.chunks_full += (int64_t)full.size();
            p.stats.chunks_partial += (int64_t)part.size();
            full.clear();
            part.clear();
            for(auto & o : occ) o = 0;
        };
        for(size_t i = 0; i < lb.bins[t].size(); i++) {
            const uint32_t id = lb.bins[t][i];
            const ListItem & it = lb.items[id];
            const int x1 = std::max(it.x1, tx0) - tx0, x2 = std::min(it.x2, tx0 + LT - 1) - tx0;
            const int y1 = std::max(it.y1, ty0) - ty0, y2 = std::min(it.y2, ty0 + LT - 1) - ty0;
            if(x2 < x1 || y2 < y1) continue;
            const int region_px = (x2 - x1 + 1) * (y2 - y1 + 1);
            if(i < first) {  // hidden by a later item that overwrites the whole tile
                p.stats.px_culled += region_px;
                continue;
            }
            if(region_px >= LIST_BIG_REGION) {
                flush();
                p.steps.push_back(make_uint4(LIST_STEP_ITEM | id, 0, 0, 0));
                p.stats.item_steps++;
                p.stats.px_item_steps += region_px;
                continue;
            }
            const uint32_t k = item_kind(it, lb.argb);
            const uint64_t rows = (y2 - y1 == 63 ? ~0ull : ((1ull << (y2 - y1 + 1)) - 1ull) << y1);
            bool conflict = false;
            for(int g = x1 >> 3; g <= (x2 >> 3); g++) conflict = conflict || (occ[g] & rows) != 0;
            if(k != kind || conflict) flush();
            kind = k;
            for(int g = x1 >> 3; g <= (x2 >> 3); g++) {
                occ[g] |= rows;
                const int lo = std::max(x1, g * 8) - g * 8, hi = std::min(x2, g * 8 + 7) - g * 8;
                const uint32_t pxmask = ((1u << (hi + 1)) - 1u) & ~((1u << lo) - 1u);
                const int ax = tx0 + g * 8 - it.x1;  // area-relative x of the group's first pixel (may be negative)
                const int ay0 = ty0 + y1 - it.y1, n_rows = y2 - y1 + 1;
                std::vector<uint4> & v = pxmask == 0xFFu ? full : part;
                const size_t at = v.size();
                v.resize(at + (size_t)n_rows);
                uint4 * o = v.data() + at;
                const uint32_t where = ((uint32_t)g << 6) | (pxmask << 9);
                int mo = ay0 * it.mask_stride + ax, so = ay0 * it.src_stride + ax;
                for(int y = y1; y <= y2; y++, mo += it.mask_stride, so += it.src_stride)
                    *o++ = make_uint4(id, (uint32_t)y | where, (uint32_t)mo, (uint32_t)so);
                p.stats.px_chunks += (int64_t)popcount8(pxmask) * n_rows;
            }
        }
        flush();
        if(all_tiles || p.steps.size() > p.bin_start[t]) {
            p.sel.push_back(t | (ow ? 0x80000000u : 0u));
            p.stats.tiles_selected++;
            if(ow) p.stats.tiles_overwritten++;
        }
    }
    p.bin_start[n_tiles] = (uint32_t)p.steps.size();
}

// The tables of a plan in one block of memory: steps | chunks | items | bin_start | sel (16-, 16-, 8-, 4-, 4-byte
// elements).  Returns the block's size; with dest != nullptr also writes it and, given the block's device address,
// fills the table pointers of `a`.
size_t pack_tables(const ListPlan & plan, const ListBuilder & lb, uint8_t * dest, const uint8_t * dev, ListArgs * a)
{
    const size_t b_steps = plan.steps.size() * sizeof(uint4), b_items = lb.items.size() * sizeof(ListItem),
                 b_chunks = plan.chunks.size() * sizeof(uint4), b_start = plan.bin_start.size() * 4, b_sel = plan.sel.size() * 4;
    if(dest) {
        uint8_t * w = dest;
        auto put = [&w](const void * p, size_t n) {
            if(n) memcpy(w, p, n);
            w += n;
        };
        put(plan.steps.data(), b_steps);
        put(plan.chunks.data(), b_chunks);
        put(lb.items.data(), b_items);
        put(plan.bin_start.data(), b_start);
        put(plan.sel.data(), b_sel);
    }
    if(a) {
        a->steps = reinterpret_cast<const uint4 *>(dev);
        a->chunks = reinterpret_cast<const uint4 *>(dev + b_steps);
        a->items = reinterpret_cast<const ListItem *>(dev + b_steps + b_chunks);
        a->bin_start = reinterpret_cast<const uint32_t *>(dev + b_steps + b_chunks + b_items);
        a->sel = reinterpret_cast<const uint32_t *>(dev + b_steps + b_chunks + b_items + b_start);
        a->n_sel = (uint32_t)plan.sel.size();
    }
    return b_steps + b_items + b_chunks + b_start + b_sel;
}

// two waves per tile pay off when the batches are large enough to keep 128 lanes busy (LVGL's row-by-row
// producers: 150+ chunks per batch); lists of many small unrelated items are better off with one
inline bool plan_is_wide(const lvgpu_list_plan_t & ps)
{
    return ps.batch_steps > 0 && (ps.chunks_full + ps.chunks_partial) >= 64 * ps.batch_steps;
}

int launch_list_kernel(const ListArgs & a, bool argb, bool wide, hipStream_t st)
{
    const uint64_t waves = (uint64_t)a.n_sel * a.n_frames;
    if(waves == 0) return LVGPU_OK;
    if(waves > 0xFFFFFFF0ull) return fail(LVGPU_ERR_ARG, "batch too large for one launch");
    if(argb) hipLaunchKernelGGL(k_blend_list_argb, dim3((uint32_t)((waves + 3) / 4)), dim3(256), 0, st, a);
    else if(wide)
        hipLaunchKernelGGL(k_blend_list<2>, dim3((uint32_t)((waves + ListCfg<2>::TPB - 1) / ListCfg<2>::TPB)),
                           dim3(ListCfg<2>::BLOCK), 0, st, a);
    else
        hipLaunchKernelGGL(k_blend_list<1>, dim3((uint32_t)((waves + ListCfg<1>::TPB - 1) / ListCfg<1>::TPB)),
                           dim3(ListCfg<1>::BLOCK), 0, st, a);
    HIP_TRY(hipGetLastError());
    return LVGPU_OK;
}

// Launch the accumulated items.  out != nullptr: also write the panel image of every tile.
int launch_list(ListBuilder & lb, uint16_t * dst, long long dst_fs, uint32_t n_frames, int swap, uint16_t * out, long long out_fs,
                int rot, int swap_out, hipStream_t st)
{
    static thread_local ListPlan plan;  // keeps its capacity between calls
    build_plan(lb, out != nullptr, plan);
    if(plan.sel.empty()) {
        lb.clear();
        return LVGPU_OK;
    }
    if((uint64_t)plan.sel.size() * n_frames > 0xFFFFFFF0ull) return fail(LVGPU_ERR_ARG, "batch too large for one launch");
    const size_t blob_bytes = pack_tables(plan, lb, nullptr, nullptr, nullptr);
    uint8_t * host = nullptr;
    int rc_stage = g_stage.acquire(blob_bytes, &host);
    if(rc_stage != LVGPU_OK) return rc_stage;
    uint8_t * dev = nullptr;
    g_alloc_calls++;
    HIP_TRY(hipMallocAsync(reinterpret_cast<void **>(&dev), blob_bytes, st));
    ListArgs a;
    memset(&a, 0, sizeof(a));
    pack_tables(plan, lb, host, dev, &a);
    const hipError_t e_copy = hipMemcpyAsync(dev, host, blob_bytes, hipMemcpyHostToDevice, st);
    if(e_copy == hipSuccess) rc_stage = g_stage.release(st);
    if(e_copy != hipSuccess || rc_stage != LVGPU_OK) {
        (void)hipFreeAsync(dev, st);
        return e_copy != hipSuccess ? fail(LVGPU_ERR_HIP, "hipMemcpyAsync: %s", hipGetErrorString(e_copy)) : rc_stage;
    }
    a.dst = dst;
    a.dst_fs = dst_fs;
    a.W = lb.W;
    a.H = lb.H;
    a.n_frames = n_frames;
    a.tiles_x = lb.tiles_x;
    a.swap = swap;
    a.out = out;
    a.out_fs = out_fs;
    a.rot = rot;
    a.swap_out = swap_out;
    const int rc = launch_list_kernel(a, lb.argb, plan_is_wide(plan.stats), st);
    const hipError_t e_free = hipFreeAsync(dev, st);  // in stream order: after the kernel has read the tables
    lb.clear();
    if(rc != LVGPU_OK) return rc;
    if(e_free != hipSuccess) return fail(LVGPU_ERR_HIP, "hipFreeAsync: %s", hipGetErrorString(e_free));
    return LVGPU_OK;
}

void add_stats(lvgpu_list_plan_t & acc, const lvgpu_list_plan_t & s)
{
    acc.tiles = s.tiles;
    acc.tiles_selected += s.tiles_selected;
    acc.tiles_overwritten += s.tiles_overwritten;
    acc.items_listed += s.items_listed;
    acc.launches++;
    acc.item_steps += s.item_steps;
    acc.batch_steps += s.batch_steps;
    acc.chunks_full += s.chunks_full;
    acc.chunks_partial += s.chunks_partial;
    acc.px_item_steps += s.px_item_steps;
    acc.px_chunks += s.px_chunks;
    acc.px_culled += s.px_culled;
}

// plan != nullptr: nothing is launched; the work the call would do is described in *plan.
int run_list(const lvgpu_blend_batch_t * geom, const lvgpu_blend_batch_t * items, int n_items, bool flush, void * out_v,
             long long out_fs, int rot, int swap_out, hipStream_t st, lvgpu_list_plan_t * plan = nullptr)
{
    if(plan) memset(plan, 0, sizeof(*plan));
    g_last_kernel = "";
    if(n_items < 0) return fail(LVGPU_ERR_ARG, "n_items < 0");
    if(n_items > 0 && items == nullptr) return fail(LVGPU_ERR_ARG, "items is NULL");
    if(geom == nullptr) return n_items == 0 && !flush ? LVGPU_OK : fail(LVGPU_ERR_ARG, "descriptor is NULL");
    if(geom->n_frames < 0) return fail(LVGPU_ERR_ARG, "n_frames < 0");
    if(flush) {
        if(out_v == nullptr) return fail(LVGPU_ERR_ARG, "out is NULL");
        if(rot < LVGPU_ROT_NONE || rot > LVGPU_ROT_270) return fail(LVGPU_ERR_ARG, "bad rotation %d", rot);
        if(geom->dst_argb8565) return fail(LVGPU_ERR_UNSUPPORTED, "flush of an ARGB8565 buffer (the reference flushes lv_color_t only)");
        if(aw(geom->buf_area) <= 0 || ah(geom->buf_area) <= 0) return fail(LVGPU_ERR_ARG, "empty buf_area");
        if((uint64_t)aw(geom->buf_area) * ah(geom->buf_area) > 0x7FFFFFFFull) return fail(LVGPU_ERR_ARG, "frame too large");
    }
    if(geom->n_frames == 0) return LVGPU_OK;
    if(geom->dst == nullptr) return fail(LVGPU_ERR_ARG, "dst is NULL");

    // resolve everything first: nothing is launched when any item is malformed
    std::vector<BlendArgs> ra((size_t)n_items);
    std::vector<lvgpu_area_t> areas((size_t)n_items);
    std::vector<int> res((size_t)n_items);
    for(int i = 0; i < n_items; i++) {
        const lvgpu_blend_batch_t & d = items[i];
        if(d.dst != geom->dst || d.dst_frame_stride != geom->dst_frame_stride || d.n_frames != geom->n_frames ||
           memcmp(&d.buf_area, &geom->buf_area, sizeof(lvgpu_area_t)) != 0 || (d.color_16_swap != 0) != (geom->color_16_swap != 0) ||
           (d.dst_argb8565 != 0) != (geom->dst_argb8565 != 0))
            return fail(LVGPU_ERR_ARG, "item %d does not address the same batch of buffers as the list", i);
        res[i] = resolve_blend(&d, ra[i], areas[i]);
        if(res[i] < 0) {
            g_err = "item " + std::to_string(i) + ": " + g_err;
            return res[i];
        }
    }
    const int W = aw(geom->buf_area), H = ah(geom->buf_area);
    uint16_t * dst = reinterpret_cast<uint16_t *>(geom->dst);
    uint16_t * out = reinterpret_cast<uint16_t *>(out_v);
    const bool argb = geom->dst_argb8565 != 0;
    // 16-byte tile rows: 8 pixels of 2 bytes, or 16 pixels of 3 bytes
    const bool tiles_ok = argb ? (W % 16) == 0 && aligned(dst, 16) && (geom->dst_frame_stride * 3) % 16 == 0
                               : (W % 8) == 0 && aligned(dst, 16) && (geom->dst_frame_stride % 8) == 0;
    const bool transposing = rot == LVGPU_ROT_90 || rot == LVGPU_ROT_270;
    const bool fuse_out = flush && tiles_ok && aligned(out, 16) && (out_fs % 8) == 0 && (!transposing || (H % 8) == 0);

    ListBuilder lb(W, H, argb);
    int rc;
    bool used_list = false;
    if(plan) {  // describe, do not draw
        static thread_local ListPlan p;
        auto close_segment = [&](bool all_tiles) {
            if(lb.items.empty() && !all_tiles) return;
            build_plan(lb, all_tiles, p);
            add_stats(*plan, p.stats);
            lb.clear();
        };
        for(int i = 0; i < n_items; i++) {
            if(res[i] != R_WORK) {
                plan->items_noop++;
                continue;
            }
            if(tiles_ok && !needs_own_pass(ra[i])) {
                lb.add(ra[i], items[i].src_format, areas[i], geom->buf_area);
                continue;
            }
            close_segment(false);
            plan->items_standalone++;
            plan->launches++;
        }
        if(fuse_out && !lb.items.empty()) {
            close_segment(true);
            plan->flush_fused = 1;
        }
        else {
            close_segment(false);
            if(flush) plan->launches++;
        }
        return LVGPU_OK;
    }
    for(int i = 0; i < n_items; i++) {
        if(res[i] != R_WORK) continue;
        if(tiles_ok && !needs_own_pass(ra[i])) {
            lb.add(ra[i], items[i].src_format, areas[i], geom->buf_area);
            continue;
        }
        if(!lb.items.empty()) {
            rc = launch_list(lb, dst, geom->dst_frame_stride, (uint32_t)geom->n_frames, geom->color_16_swap ? 1 : 0, nullptr, 0,
                             0, 0, st);
            if(rc != LVGPU_OK) return rc;
            used_list = true;
        }
        rc = launch_blend(ra[i], items[i].src_format, st, argb);
        if(rc != LVGPU_OK) return rc;
    }
    if(fuse_out && !lb.items.empty()) {
        rc = launch_list(lb, dst, geom->dst_frame_stride, (uint32_t)geom->n_frames, geom->color_16_swap ? 1 : 0, out, out_fs, rot,
                         swap_out ? 1 : 0, st);
        if(rc == LVGPU_OK) g_last_kernel = "k_blend_list(fused)";
        return rc;
    }
    if(!lb.items.empty()) {
        rc = launch_list(lb, dst, geom->dst_frame_stride, (uint32_t)geom->n_frames, geom->color_16_swap ? 1 : 0, nullptr, 0, 0, 0,
                         st);
        if(rc != LVGPU_OK) return rc;
        used_list = true;
    }
    if(flush) return launch_rotate(dst, geom->dst_frame_stride, out, out_fs, W, H, rot, swap_out, geom->n_frames, st);
    if(used_list) g_last_kernel = "k_blend_list";
    return LVGPU_OK;
}

// A display list planned once and kept on the device (lvgpu_list_create / _replay / _destroy).
struct CompiledList {
    uint32_t magic;
    int device;
    uint8_t * dev;   // the tables
    ListArgs a;      // everything but dst / n_frames / out / out_fs, which a replay may set
    bool argb, wide, flush;
    bool per_frame_inputs;  // some item reads a source or mask that advances from frame to frame
    void * dst;      // as recorded
    int n_frames;
};
constexpr uint32_t LIST_MAGIC = 0x4C4C5354u;  // "LLST"

}  // namespace

extern "C" {

int lvgpu_version(void) { return LVGPU_VERSION; }
const char * lvgpu_last_error(void) { return g_err.c_str(); }
const char * lvgpu_last_kernel(void) { return g_last_kernel; }

int lvgpu_device_count(void)
{
    int n = 0;
    if(hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int lvgpu_init(int device)
{
    int n = 0;
    if(hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(LVGPU_ERR_NO_DEVICE, "no HIP device visible");
    if(device < 0 || device >= n) return fail(LVGPU_ERR_ARG, "device %d out of range (0..%d)", device, n - 1);
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if(strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(LVGPU_ERR_NO_DEVICE, "device %d is %s; this library carries gfx950 code only", device, prop.gcnArchName);
    HIP_TRY(hipFree(nullptr));
    uint32_t * p = nullptr;
    return scratch_on_current_device(&p);  // so that lvgpu_blend_batch_noalloc finds it without a lookup of its own
}

int lvgpu_malloc(void ** p, size_t bytes)
{
    if(!p) return fail(LVGPU_ERR_ARG, "NULL out pointer");
    g_alloc_calls++;
    HIP_TRY(hipMalloc(p, bytes));
    return LVGPU_OK;
}
int lvgpu_free(void * p)
{
    HIP_TRY(hipFree(p));
    return LVGPU_OK;
}
int lvgpu_memcpy_h2d(void * d, const void * s, size_t n, void * stream)
{
    HIP_TRY(hipMemcpyAsync(d, s, n, hipMemcpyHostToDevice, (hipStream_t)stream));
    return LVGPU_OK;
}
int lvgpu_memcpy_d2h(void * d, const void * s, size_t n, void * stream)
{
    HIP_TRY(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToHost, (hipStream_t)stream));
    return LVGPU_OK;
}
int lvgpu_memset(void * d, int v, size_t n, void * stream)
{
    HIP_TRY(hipMemsetAsync(d, v, n, (hipStream_t)stream));
    return LVGPU_OK;
}
int lvgpu_stream_synchronize(void * stream)
{
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return LVGPU_OK;
}

int lvgpu_blend_batch(const lvgpu_blend_batch_t * d, void * stream)
{
    BlendArgs a;
    lvgpu_area_t area;
    g_last_kernel = "";
    int r = resolve_blend(d, a, area);
    if(r <= 0) return r;
    return launch_blend(a, d->src_format, (hipStream_t)stream, d->dst_argb8565 != 0);
}

int lvgpu_blend_batch_noalloc(const lvgpu_blend_batch_t * d, void * stream)
{
    BlendArgs a;
    lvgpu_area_t area;
    g_last_kernel = "";
    if(d != nullptr && d->n_frames > LVGPU_NOALLOC_MAX_FRAMES)
        return fail(LVGPU_ERR_ARG, "lvgpu_blend_batch_noalloc takes at most %d frames per call (%d given); split the batch or use "
                    "lvgpu_blend_batch", LVGPU_NOALLOC_MAX_FRAMES, d->n_frames);
    int r = resolve_blend(d, a, area);
    if(r <= 0) return r;
    uint32_t * buf = nullptr;
    r = scratch_on_current_device(&buf);
    if(r != LVGPU_OK) return r;
    return launch_blend(a, d->src_format, (hipStream_t)stream, d->dst_argb8565 != 0, buf);
}

int lvgpu_flush_rotate_batch(const lvgpu_flush_batch_t * d, void * stream)
{
    g_last_kernel = "";
    if(d == nullptr) return fail(LVGPU_ERR_ARG, "descriptor is NULL");
    if(d->n_frames < 0 || d->width <= 0 || d->height <= 0) return fail(LVGPU_ERR_ARG, "bad geometry");
    if(d->rotation < LVGPU_ROT_NONE || d->rotation > LVGPU_ROT_270) return fail(LVGPU_ERR_ARG, "bad rotation %d", d->rotation);
    if(d->n_frames == 0) return LVGPU_OK;
    if(d->in == nullptr || d->out == nullptr) return fail(LVGPU_ERR_ARG, "NULL buffer");
    if((uint64_t)d->width * d->height > 0x7FFFFFFFull) return fail(LVGPU_ERR_ARG, "frame too large");
    return launch_rotate(reinterpret_cast<const uint16_t *>(d->in), d->in_frame_stride, reinterpret_cast<uint16_t *>(d->out),
                         d->out_frame_stride, d->width, d->height, d->rotation, d->swap_bytes, d->n_frames, (hipStream_t)stream);
}

int lvgpu_blend_flush_batch(const lvgpu_blend_flush_batch_t * d, void * stream)
{
    g_last_kernel = "";
    if(d == nullptr) return fail(LVGPU_ERR_ARG, "descriptor is NULL");
    if(d->out == nullptr) return fail(LVGPU_ERR_ARG, "out is NULL");
    if(d->rotation < LVGPU_ROT_NONE || d->rotation > LVGPU_ROT_270) return fail(LVGPU_ERR_ARG, "bad rotation %d", d->rotation);
    if(d->blend.dst_argb8565) return fail(LVGPU_ERR_UNSUPPORTED, "flush of an ARGB8565 buffer (the reference flushes lv_color_t only)");
    hipStream_t st = (hipStream_t)stream;
    BlendArgs a;
    lvgpu_area_t area;
    int r = resolve_blend(&d->blend, a, area, LVGPU_BLEND_FLAG_DISCARD_DST);
    if(r < 0) return r;
    const lvgpu_blend_batch_t & b = d->blend;
    const int W = aw(b.buf_area), H = ah(b.buf_area);
    if(b.n_frames == 0) return LVGPU_OK;
    const bool transposing = d->rotation == LVGPU_ROT_90 || d->rotation == LVGPU_ROT_270;
    const bool out_ok = aligned(d->out, 16) && (d->out_frame_stride % 8) == 0;
    if(r == R_WORK && (out_ok || transposing) && a.w == W && a.h == H) {  // the blend touches the whole buffer: one pass over memory
        a.out = reinterpret_cast<uint16_t *>(d->out);
        a.out_fs = d->out_frame_stride;
        a.rot = d->rotation;
        a.swap_out = d->swap_bytes ? 1 : 0;
        // (anti-aliasing off and a quarter turn: two passes -- the quarter-turn kernels are compiled without that switch)
        const bool ok = out_ok && !(transposing && a.no_aa) && (transposing ? tile_ok(a, b.src_format) : linear_ok(a, b.src_format));
        // a quarter turn of a frame off the 8-pixel grid, or of misaligned buffers: still one pass (k_tile_any); the
        // blend must cover the whole buffer row by row (dst_stride == W holds, source and mask keep their own strides)
        const bool any = transposing && !ok && !a.no_aa;
        if(ok || any) {
            a.discard_dst = (b.flags & LVGPU_BLEND_FLAG_DISCARD_DST) ? 1 : 0;
            uint32_t * owned = nullptr;
            int rc = prepare_first_nb(a, st, &owned);
            if(rc != LVGPU_OK) return rc;
            if(any) {
                g_last_kernel = "k_tile_any(fused)";
                rc = launch_tile_any(b.src_format, a, st);
            }
            else if(transposing && (a.h % 32) == 0 && a.w <= BAND_MAX_W && a.mask && a.opa_mode == OM_MASK && a.blend_fn == BF_NORMAL &&
                    (b.src_format == SRC_FILL || b.src_format == SRC_RGB565) && !band_disabled()) {
                // a mask plane beside the pixels, mixed at full opacity: bands of whole rows read it in whole lines
                // (see k_band; with an opacity below the threshold or a blend mode the tile kernel is 10 points ahead)
                g_last_kernel = "k_band(fused)";
                rc = launch_band(b.src_format, a, st);
            }
            else if(transposing && wants_col(a)) {
                g_last_kernel = "k_tile_col(fused)";
                rc = launch_tile_col(b.src_format, a, st);
            }
            else if(transposing) {
                g_last_kernel = "k_tile(fused)";
                rc = launch_tile(b.src_format, a, st);
            }
            else {
                g_last_kernel = "k_blend_linear(fused)";
                rc = dispatch_src<LaunchLinear>(b.src_format, a, st);
            }
            if(owned) {
                const hipError_t e_free = hipFreeAsync(owned, st);
                if(rc == LVGPU_OK && e_free != hipSuccess) return fail(LVGPU_ERR_HIP, "hipFreeAsync: %s", hipGetErrorString(e_free));
            }
            return rc;
        }
        a.out = nullptr;
        a.discard_dst = 0;
    }
    if(r == R_WORK) {
        int rc = launch_blend(a, b.src_format, st);
        if(rc != LVGPU_OK) return rc;
    }
    return launch_rotate(reinterpret_cast<const uint16_t *>(b.dst), b.dst_frame_stride, reinterpret_cast<uint16_t *>(d->out),
                         d->out_frame_stride, W, H, d->rotation, d->swap_bytes, b.n_frames, st);
}

int lvgpu_blend_list_batch(const lvgpu_blend_batch_t * items, int32_t n_items, void * stream)
{
    return run_list(n_items > 0 ? items : nullptr, items, n_items, false, nullptr, 0, 0, 0, (hipStream_t)stream);
}

int lvgpu_blend_list_plan(const lvgpu_blend_batch_t * geom, const lvgpu_blend_batch_t * items, int32_t n_items, int32_t flush,
                          int32_t rotation, lvgpu_list_plan_t * plan)
{
    if(plan == nullptr) return fail(LVGPU_ERR_ARG, "plan is NULL");
    // the output pointer is only inspected for its alignment: a fused flush is assumed possible
    return run_list(geom ? geom : (n_items > 0 ? items : nullptr), items, n_items, flush != 0, flush ? (void *)(uintptr_t)256 : nullptr, 0,
                    rotation, 0, nullptr, plan);
}

int lvgpu_blend_list_flush_batch(const lvgpu_blend_batch_t * geom, const lvgpu_blend_batch_t * items, int32_t n_items, void * out,
                                 int64_t out_frame_stride, int32_t rotation, int32_t swap_bytes, void * stream)
{
    return run_list(geom, items, n_items, true, out, out_frame_stride, rotation, swap_bytes, (hipStream_t)stream);
}

int lvgpu_list_create(const lvgpu_blend_batch_t * geom, const lvgpu_blend_batch_t * items, int32_t n_items, int32_t flush,
                      int32_t rotation, int32_t swap_bytes, lvgpu_list_t ** out_list)
{
    g_last_kernel = "";
    if(out_list == nullptr) return fail(LVGPU_ERR_ARG, "NULL out pointer");
    *out_list = nullptr;
    if(n_items < 0 || (n_items > 0 && items == nullptr)) return fail(LVGPU_ERR_ARG, "bad item list");
    if(geom == nullptr) geom = n_items > 0 ? items : nullptr;
    if(geom == nullptr) return fail(LVGPU_ERR_ARG, "descriptor is NULL");
    if(geom->n_frames <= 0 || geom->dst == nullptr) return fail(LVGPU_ERR_ARG, "geometry needs dst and n_frames > 0");
    if(flush && (rotation < LVGPU_ROT_NONE || rotation > LVGPU_ROT_270)) return fail(LVGPU_ERR_ARG, "bad rotation %d", rotation);
    if(aw(geom->buf_area) <= 0 || ah(geom->buf_area) <= 0) return fail(LVGPU_ERR_ARG, "empty buf_area");
    const int W = aw(geom->buf_area), H = ah(geom->buf_area);
    const bool argb = geom->dst_argb8565 != 0;
    if(flush && argb) return fail(LVGPU_ERR_UNSUPPORTED, "flush of an ARGB8565 buffer (the reference flushes lv_color_t only)");
    const bool tiles_ok = argb ? (W % 16) == 0 && aligned(geom->dst, 16) && (geom->dst_frame_stride * 3) % 16 == 0
                               : (W % 8) == 0 && aligned(geom->dst, 16) && (geom->dst_frame_stride % 8) == 0;
    const bool transposing = rotation == LVGPU_ROT_90 || rotation == LVGPU_ROT_270;
    if(!tiles_ok || (flush && transposing && (H % 8) != 0))
        return fail(LVGPU_ERR_UNSUPPORTED, "this buffer shape does not take the tile pass; use lvgpu_blend_list_batch");
    ListBuilder lb(W, H, argb);
    for(int i = 0; i < n_items; i++) {
        const lvgpu_blend_batch_t & d = items[i];
        if(d.dst != geom->dst || d.dst_frame_stride != geom->dst_frame_stride || d.n_frames != geom->n_frames ||
           memcmp(&d.buf_area, &geom->buf_area, sizeof(lvgpu_area_t)) != 0 || (d.color_16_swap != 0) != (geom->color_16_swap != 0) ||
           (d.dst_argb8565 != 0) != argb)
            return fail(LVGPU_ERR_ARG, "item %d does not address the same batch of buffers as the list", i);
        BlendArgs ra;
        lvgpu_area_t area;
        int r = resolve_blend(&d, ra, area);
        if(r < 0) {
            g_err = "item " + std::to_string(i) + ": " + g_err;
            return r;
        }
        if(r != R_WORK) continue;
        if(needs_scan_order(ra))
            return fail(LVGPU_ERR_UNSUPPORTED, "item %d needs a pass over its whole area first (translucent fill whose result "
                        "on black depends on scan order); use lvgpu_blend_list_batch", i);
        if(ra.no_aa)
            return fail(LVGPU_ERR_UNSUPPORTED, "item %d has LVGPU_BLEND_FLAG_NO_AA, which the tile pass does not take; use "
                        "lvgpu_blend_list_batch", i);
        lb.add(ra, d.src_format, area, geom->buf_area);
    }
    ListPlan plan;
    build_plan(lb, flush != 0, plan);
    CompiledList * cl = new CompiledList();
    memset(cl, 0, sizeof(*cl));
    cl->magic = LIST_MAGIC;
    const size_t bytes = pack_tables(plan, lb, nullptr, nullptr, nullptr);
    std::vector<uint8_t> host(bytes);
    hipError_t e = hipGetDevice(&cl->device);
    g_alloc_calls++;
    if(e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&cl->dev), bytes ? bytes : 16);
    if(e == hipSuccess) {
        pack_tables(plan, lb, host.data(), cl->dev, &cl->a);
        e = hipMemcpy(cl->dev, host.data(), bytes, hipMemcpyHostToDevice);
    }
    if(e != hipSuccess) {
        if(cl->dev) (void)hipFree(cl->dev);
        delete cl;
        return fail(LVGPU_ERR_HIP, "lvgpu_list_create: %s", hipGetErrorString(e));
    }
    cl->a.dst_fs = geom->dst_frame_stride;
    cl->a.W = W;
    cl->a.H = H;
    cl->a.tiles_x = lb.tiles_x;
    cl->a.swap = geom->color_16_swap ? 1 : 0;
    cl->a.rot = rotation;
    cl->a.swap_out = swap_bytes ? 1 : 0;
    cl->argb = argb;
    cl->wide = plan_is_wide(plan.stats);
    cl->flush = flush != 0;
    cl->dst = geom->dst;
    cl->n_frames = geom->n_frames;
    for(const ListItem & it : lb.items) cl->per_frame_inputs = cl->per_frame_inputs || it.src_fs != 0 || it.mask_fs != 0;
    *out_list = reinterpret_cast<lvgpu_list_t *>(cl);
    return LVGPU_OK;
}

int lvgpu_list_replay(const lvgpu_list_t * list, void * dst, int32_t n_frames, void * out, int64_t out_frame_stride, void * stream)
{
    g_last_kernel = "";
    const CompiledList * cl = reinterpret_cast<const CompiledList *>(list);
    if(cl == nullptr || cl->magic != LIST_MAGIC) return fail(LVGPU_ERR_ARG, "not a list handle");
    if(n_frames < 0) return fail(LVGPU_ERR_ARG, "n_frames < 0");
    int cur = -1;
    HIP_TRY(hipGetDevice(&cur));
    if(cur != cl->device) return fail(LVGPU_ERR_ARG, "the list lives on device %d, the current device is %d", cl->device, cur);
    ListArgs a = cl->a;
    a.dst = reinterpret_cast<uint16_t *>(dst ? dst : cl->dst);
    a.n_frames = (uint32_t)(n_frames ? n_frames : cl->n_frames);
    if(cl->per_frame_inputs && a.n_frames > (uint32_t)cl->n_frames)
        return fail(LVGPU_ERR_ARG, "the list reads per-frame sources or masks recorded for %d frames; %u frames would read past them",
                    cl->n_frames, a.n_frames);
    if(!aligned(a.dst, 16)) return fail(LVGPU_ERR_ARG, "dst must be 16-byte aligned");
    if(cl->flush) {
        if(out == nullptr) return fail(LVGPU_ERR_ARG, "out is NULL");
        if(!aligned(out, 16) || (out_frame_stride % 8) != 0) return fail(LVGPU_ERR_ARG, "out must be 16-byte aligned, its stride a multiple of 8");
        a.out = reinterpret_cast<uint16_t *>(out);
        a.out_fs = out_frame_stride;
    }
    int rc = launch_list_kernel(a, cl->argb, cl->wide, (hipStream_t)stream);
    if(rc == LVGPU_OK) g_last_kernel = cl->flush ? "k_blend_list(fused)" : "k_blend_list";
    return rc;
}

int lvgpu_list_destroy(lvgpu_list_t * list)
{
    CompiledList * cl = reinterpret_cast<CompiledList *>(list);
    if(cl == nullptr) return LVGPU_OK;
    if(cl->magic != LIST_MAGIC) return fail(LVGPU_ERR_ARG, "not a list handle");
    cl->magic = 0;
    hipError_t e = hipFree(cl->dev);
    delete cl;
    if(e != hipSuccess) return fail(LVGPU_ERR_HIP, "hipFree: %s", hipGetErrorString(e));
    return LVGPU_OK;
}

int lvgpu_flush_chunks(const lvgpu_area_t * area, int32_t hor_res, int32_t ver_res, int32_t rot, uint32_t trans_size,
                       lvgpu_flush_chunk_t * out, int32_t max_chunks)
{
    if(area == nullptr) return fail(LVGPU_ERR_ARG, "area is NULL");
    if(rot < LVGPU_ROT_NONE || rot > LVGPU_ROT_270) return fail(LVGPU_ERR_ARG, "bad rotation %d", rot);
    const int x_start = area->x1, x_end = area->x2, y_start = area->y1, y_end = area->y2;
    const int W = x_end - x_start + 1, H = y_end - y_start + 1;
    if(W <= 0 || H <= 0) return fail(LVGPU_ERR_ARG, "empty area");
    const bool transposing = rot == LVGPU_ROT_90 || rot == LVGPU_ROT_270;
    const int step = transposing ? (int)(trans_size / (uint32_t)H) : (int)(trans_size / (uint32_t)W);
    if(step <= 0) return fail(LVGPU_ERR_ARG, "trans_size %u is smaller than one line of the flushed area", trans_size);
    const int span = transposing ? W : H;  // chunks advance along x (rotated) or y
    const int line = transposing ? H : W;  // pixels per advanced line in the panel image
    const int max_step = step > span ? span : step;
    int n_chunks = 0;
    for(int done = 0; done < span; done += max_step, n_chunks++) {
        if(out == nullptr || n_chunks >= max_chunks) continue;
        const int n = span - done > max_step ? max_step : span - done;
        lvgpu_flush_chunk_t & c = out[n_chunks];
        int xd0, yd0, xd1, yd1;
        switch(rot) {
            case LVGPU_ROT_90: {  // columns from the left edge (:482-484, :506-509)
                const int xs = x_start + done, xe = xs + n - 1;
                xd0 = ver_res - y_end - 1;
                xd1 = ver_res - y_start - 1;
                yd0 = xs;
                yd1 = xe;
                break;
            }
            case LVGPU_ROT_270: {  // columns from the right edge (:485-487, :517-520)
                const int xe = x_end - done, xs = xe - n + 1;
                xd0 = y_start;
                xd1 = y_end;
                yd0 = hor_res - xe - 1;
                yd1 = hor_res - xs - 1;
                break;
            }
            case LVGPU_ROT_180: {  // rows from the bottom edge (:491-494, :528-531)
                const int ye = y_end - done, ys = ye - n + 1;
                xd0 = hor_res - x_end - 1;
                xd1 = hor_res - x_start - 1;
                yd0 = ver_res - ye - 1;
                yd1 = ver_res - ys - 1;
                break;
            }
            default: {  // rows from the top edge (:488-490, :539-542)
                const int ys = y_start + done, ye = ys + n - 1;
                xd0 = x_start;
                xd1 = x_end;
                yd0 = ys;
                yd1 = ye;
                break;
            }
        }
        c.x_start = xd0;
        c.y_start = yd0;
        c.x_end = xd1 + 1;
        c.y_end = yd1 + 1;
        c.offset_px = (int64_t)done * line;
        c.n_px = n * line;
        c.reserved = 0;
    }
    return n_chunks;
}

int lvgpu_axs15231b_transactions(const lvgpu_flush_chunk_t * chunk, int32_t x_gap, int32_t y_gap, int32_t qspi,
                                 int32_t bits_per_pixel, lvgpu_panel_txn_t out[3])
{
    if(chunk == nullptr || out == nullptr) return fail(LVGPU_ERR_ARG, "NULL argument");
    if(bits_per_pixel != 16 && bits_per_pixel != 24) return fail(LVGPU_ERR_ARG, "bits_per_pixel %d (16 or 24)", bits_per_pixel);
    if(!(chunk->x_start < chunk->x_end && chunk->y_start < chunk->y_end))  // the driver asserts this (:297-298)
        return fail(LVGPU_ERR_ARG, "start position must be smaller than end position");
    const int xs = chunk->x_start + x_gap, xe = chunk->x_end + x_gap;
    const int ys = chunk->y_start + y_gap, ye = chunk->y_end + y_gap;
    auto window = [&](lvgpu_panel_txn_t & t, uint32_t cmd, int from, int to_excl) {
        t = lvgpu_panel_txn_t{};
        t.cmd = qspi ? (0x02u << 24) | (cmd << 8) : cmd;
        t.param_len = 4;
        t.param[0] = (uint8_t)((from >> 8) & 0xFF);
        t.param[1] = (uint8_t)(from & 0xFF);
        t.param[2] = (uint8_t)(((to_excl - 1) >> 8) & 0xFF);
        t.param[3] = (uint8_t)((to_excl - 1) & 0xFF);
    };
    int n = 0;
    window(out[n++], LVGPU_LCD_CMD_CASET, xs, xe);
    if(!qspi) window(out[n++], LVGPU_LCD_CMD_RASET, ys, ye);
    lvgpu_panel_txn_t & c = out[n++];
    c = lvgpu_panel_txn_t{};
    const uint32_t cmd = ys == 0 ? LVGPU_LCD_CMD_RAMWR : LVGPU_LCD_CMD_RAMWRC;
    c.cmd = qspi ? (0x32u << 24) | (cmd << 8) : cmd;
    c.is_color = 1;
    c.data_offset_px = chunk->offset_px;
    c.data_len = (int64_t)(xe - xs) * (ye - ys) * bits_per_pixel / 8;
    return n;
}

int lvgpu_color_from_rgb888(const uint32_t * in, uint16_t * out, size_t n, int swap, void * stream)
{
    if(n == 0) return LVGPU_OK;
    if(!in || !out) return fail(LVGPU_ERR_ARG, "NULL buffer");
    size_t done = 0;
    if(aligned(in, 16) && aligned(out, 16) && n >= 8) {
        const size_t n8 = n / 8;
        hipLaunchKernelGGL(k_from_rgb888_v, vector_grid(n8), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const uint4 *>(in),
                           reinterpret_cast<uint4 *>(out), n8, swap ? 1 : 0);
        HIP_TRY(hipGetLastError());
        done = n8 * 8;
    }
    if(done < n) {
        hipLaunchKernelGGL(k_from_rgb888, stream_grid(n - done), dim3(256), 0, (hipStream_t)stream, in + done, out + done, n - done, swap ? 1 : 0);
        HIP_TRY(hipGetLastError());
    }
    return LVGPU_OK;
}
int lvgpu_color_to32(const uint16_t * in, uint32_t * out, size_t n, int swap, void * stream)
{
    if(n == 0) return LVGPU_OK;
    if(!in || !out) return fail(LVGPU_ERR_ARG, "NULL buffer");
    size_t done = 0;
    if(aligned(in, 16) && aligned(out, 16) && n >= 8) {
        const size_t n8 = n / 8;
        hipLaunchKernelGGL(k_to32_v, vector_grid(n8), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const uint4 *>(in),
                           reinterpret_cast<uint4 *>(out), n8, swap ? 1 : 0);
        HIP_TRY(hipGetLastError());
        done = n8 * 8;
    }
    if(done < n) {
        hipLaunchKernelGGL(k_to32, stream_grid(n - done), dim3(256), 0, (hipStream_t)stream, in + done, out + done, n - done, swap ? 1 : 0);
        HIP_TRY(hipGetLastError());
    }
    return LVGPU_OK;
}
int lvgpu_image_split(const void * src, int fmt, uint16_t * color, uint8_t * alpha, size_t n, int swap, void * stream)
{
    if(n == 0) return LVGPU_OK;
    if(!src || !color || !alpha) return fail(LVGPU_ERR_ARG, "NULL buffer");
    if(fmt < LVGPU_SRC_ARGB8888 || fmt > LVGPU_SRC_ARGB8565) return fail(LVGPU_ERR_ARG, "bad src_format %d", fmt);
    return lvgpu_image_convert(src, fmt, color, alpha, n, swap, nullptr, stream);  // no key, no recolor: the plain split
}
int lvgpu_image_convert(const void * src, int fmt, uint16_t * color, uint8_t * alpha, size_t n, int swap,
                        const lvgpu_image_opts_t * opts, void * stream)
{
    if(n == 0) return LVGPU_OK;
    if(!src || !color || !alpha) return fail(LVGPU_ERR_ARG, "NULL buffer");
    if(fmt < LVGPU_SRC_RGB565 || fmt > LVGPU_SRC_ARGB8565) return fail(LVGPU_ERR_ARG, "bad src_format %d", fmt);
    lvgpu_image_opts_t o;
    memset(&o, 0, sizeof(o));
    if(opts) o = *opts;
    PremultFill rc;
    memset(&rc, 0, sizeof(rc));
    const int recolor = o.recolor_opa > 2 ? 1 : 0;  // LV_OPA_MIN, lv_draw_sw_img.c:163
    if(recolor) {                                   // lv_color_premult, lv_color.h:474-488
        const uint32_t c = swap ? bswap16(o.recolor) : o.recolor;
        rc.pm_r = (c >> 11) * o.recolor_opa;
        rc.pm_g = ((c >> 5) & 0x3Fu) * o.recolor_opa;
        rc.pm_b = (c & 0x1Fu) * o.recolor_opa;
        rc.opa_inv = 255u - o.recolor_opa;
    }
    size_t done = 0;
    const size_t src_al = fmt == LVGPU_SRC_ARGB8565 ? 8 : 16, bpp = (size_t)src_bytes_per_px(fmt);
    if(aligned(src, src_al) && aligned(color, 16) && aligned(alpha, 8) && n >= 8) {
        const size_t n8 = n / 8;
        const uint32_t key_native = swap ? bswap16(o.chroma_key) : o.chroma_key;
        hipLaunchKernelGGL(vector_convert_kernel(fmt), vector_grid(n8), dim3(256), 0, (hipStream_t)stream, src, reinterpret_cast<uint4 *>(color),
                           reinterpret_cast<uint2 *>(alpha), n8, swap ? 1 : 0, o.chroma_keyed ? 1 : 0, key_native, recolor, rc);
        HIP_TRY(hipGetLastError());
        done = n8 * 8;
    }
    if(done < n) {
        hipLaunchKernelGGL(k_image_convert, stream_grid(n - done), dim3(256), 0, (hipStream_t)stream,
                           reinterpret_cast<const uint8_t *>(src) + done * bpp, fmt, color + done, alpha + done, n - done, swap ? 1 : 0,
                           o.chroma_keyed ? 1 : 0, (uint32_t)o.chroma_key, recolor, rc);
        HIP_TRY(hipGetLastError());
    }
    return LVGPU_OK;
}
int lvgpu_color_fill(uint16_t * buf, uint16_t color, size_t n, void * stream)
{
    if(n == 0) return LVGPU_OK;
    if(!buf) return fail(LVGPU_ERR_ARG, "NULL buffer");
    if(!aligned(buf, 2)) return fail(LVGPU_ERR_ARG, "buffer of lv_color_t not 2-byte aligned");
    const uint32_t first_px = (uint32_t)((reinterpret_cast<uintptr_t>(buf) & 15u) >> 1);
    const size_t n_words = (first_px + n + 7) / 8;
    // one word per lane, the blocks of an XCD on one contiguous eighth (a capped grid would stride through the buffer)
    const size_t blocks = std::min<size_t>((n_words + 255) / 256, 0x7FFFFFFFull);
    hipLaunchKernelGGL(k_fill16, dim3((uint32_t)blocks), dim3(256), 0, (hipStream_t)stream, buf - first_px, first_px, color, n, n_words);
    HIP_TRY(hipGetLastError());
    return LVGPU_OK;
}
int lvgpu_buffer_copy_batch(void * dst, int64_t dst_fs, int32_t dst_stride, lvgpu_area_t da, const void * src, int64_t src_fs,
                            int32_t src_stride, lvgpu_area_t sa, int32_t n_frames, void * stream)
{
    if(n_frames == 0) return LVGPU_OK;
    if(!dst || !src || n_frames < 0) return fail(LVGPU_ERR_ARG, "bad arguments");
    const int w = aw(da), h = ah(da);  // lv_draw_sw.c:97-99: width and row count come from dest_area
    if(w <= 0 || h <= 0) return LVGPU_OK;
    // a copy is an opaque map: the same kernels, the same rules about what is read and written (lv_draw_sw.c:91-99)
    BlendArgs a;
    memset(&a, 0, sizeof(a));
    a.dst = reinterpret_cast<uint16_t *>(dst) + (size_t)dst_stride * da.y1 + da.x1;                 // :91-92
    a.dst_fs = dst_fs;
    a.dst_stride = dst_stride;
    a.src = reinterpret_cast<const uint16_t *>(src) + (size_t)src_stride * sa.y1 + sa.x1;           // :94-95
    a.src_fs = src_fs;
    a.src_stride = src_stride;
    a.w = w;
    a.h = h;
    a.n_frames = n_frames;
    a.opa = 255;
    a.raw_opa = 255;
    a.opa_mode = OM_CONST;
    a.blend_fn = BF_NORMAL;
    return launch_blend(a, SRC_RGB565, (hipStream_t)stream);
}

int lvgpu_synth(void * out, int kind, uint64_t seed, uint64_t first, size_t n, void * stream)
{
    if(n == 0) return LVGPU_OK;
    if(!out || kind < 0 || kind > 4) return fail(LVGPU_ERR_ARG, "bad arguments");
    hipLaunchKernelGGL(k_synth, stream_grid(n), dim3(256), 0, (hipStream_t)stream, out, kind, seed, first, n);
    HIP_TRY(hipGetLastError());
    return LVGPU_OK;
}
int lvgpu_checksum(const void * in, int elem_bytes, uint64_t first, size_t n, uint64_t * sum, void * stream)
{
    if(n == 0) return LVGPU_OK;
    if(!in || !sum || (elem_bytes != 1 && elem_bytes != 2 && elem_bytes != 4)) return fail(LVGPU_ERR_ARG, "bad arguments");
    hipLaunchKernelGGL(k_checksum, stream_grid(n), dim3(256), 0, (hipStream_t)stream, in, elem_bytes, first, n,
                       reinterpret_cast<unsigned long long *>(sum));
    HIP_TRY(hipGetLastError());
    return LVGPU_OK;
}

void lvgpu_debug_counters(lvgpu_debug_counters_t * out)
{
    if(out == nullptr) return;
    out->alloc_calls = g_alloc_calls;
    out->first_nonblack_launches = g_first_nb_launches;
    out->band_pad_bytes = band_pad_of_last_launch();
}

int lvgpu_stream_create_cu_mask(void ** stream, const uint32_t * mask_words, int n_words)
{
    if(!stream || !mask_words || n_words <= 0) return fail(LVGPU_ERR_ARG, "bad arguments");
    hipStream_t st;
    HIP_TRY(hipExtStreamCreateWithCUMask(&st, (uint32_t)n_words, mask_words));
    *stream = st;
    return LVGPU_OK;
}
int lvgpu_stream_create(void ** stream)
{
    if(!stream) return fail(LVGPU_ERR_ARG, "NULL out pointer");
    hipStream_t st;
    HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    *stream = st;
    return LVGPU_OK;
}
int lvgpu_stream_destroy(void * stream)
{
    HIP_TRY(hipStreamDestroy((hipStream_t)stream));
    return LVGPU_OK;
}
int lvgpu_xcc_census(uint32_t * dev_counts8, int n_blocks, void * stream)
{
    if(!dev_counts8 || n_blocks <= 0) return fail(LVGPU_ERR_ARG, "bad arguments");
    hipLaunchKernelGGL(k_xcc_census, dim3(n_blocks), dim3(64), 0, (hipStream_t)stream, dev_counts8);
    HIP_TRY(hipGetLastError());
    return LVGPU_OK;
}

}  // extern "C"
