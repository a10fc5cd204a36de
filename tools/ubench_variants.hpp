// Candidate kernels for the fused blend + rotate path (development only).
#pragma once

// ---- V1: k_tile with a configurable lane grid LX x LY (LX*LY = 64), tile = 8LX x 8LY pixels ---------------------
template <int SRC, int LX, bool NT, int REMAP>
__global__ __launch_bounds__(256) void k_tile_v1(BlendArgs a, int W, int H, uint32_t tiles_x, uint32_t tpf, uint32_t n_tiles,
                                                 uint32_t n_blocks)
{
    constexpr int LY = 64 / LX;
    constexpr int TX = 8 * LX, TY = 8 * LY;
    __shared__ uint4 lds[4][512];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    uint32_t blk = blockIdx.x;
    if(REMAP) {  // blocks that share an XCD (b % 8) get a contiguous range of the batch
        const uint32_t q = n_blocks / 8, r = n_blocks % 8, x = blk % 8;
        blk = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + blk / 8;
    }
    const uint32_t wt = __builtin_amdgcn_readfirstlane(blk * 4 + wave);
    const bool live = wt < n_tiles;
    const uint32_t f = live ? wt / tpf : 0;
    const uint32_t t = live ? wt - f * tpf : 0;
    const uint32_t ty = t / tiles_x, tx = t - ty * tiles_x;
    constexpr bool BLEND = SRC != SRC_COPY;

    const int bx = lane % LX, by = lane / LX;
    const int x0 = tx * TX + bx * 8, y0 = ty * TY + by * 8;
    const bool inside = live && x0 < W && y0 < H;

    uint4 r[8];
    if(inside) {
        uint16_t * dst = a.dst + (long long)f * a.dst_fs;
        const long long src_px0 = (long long)f * a.src_fs;
        uint4 d[8];
        uint32_t sw[8][SrcWords<SRC>::N];
#pragma unroll
        for(int i = 0; i < 8; i++) {
            const size_t p = (size_t)(y0 + i) * W + x0;
            d[i] = NT ? nt_load(reinterpret_cast<const uint4 *>(dst + p)) : *reinterpret_cast<const uint4 *>(dst + p);
            if(BLEND && SRC != SRC_FILL) {
                if(NT && (SRC == SRC_ARGB8888)) {
                    const uint4 * pp = reinterpret_cast<const uint4 *>(reinterpret_cast<const uint32_t *>(a.src) + src_px0 + p);
                    uint4 v0 = nt_load(pp), v1 = nt_load(pp + 1);
                    sw[i][0] = v0.x; sw[i][1] = v0.y; sw[i][2] = v0.z; sw[i][3] = v0.w;
                    sw[i][4] = v1.x; sw[i][5] = v1.y; sw[i][6] = v1.z; sw[i][7] = v1.w;
                }
                else load_src<SRC>(a.src, src_px0 + (long long)p, sw[i]);
            }
        }
#pragma unroll
        for(int i = 0; i < 8; i++) {
            r[i] = d[i];
            if(BLEND) {
                const size_t p = (size_t)(y0 + i) * W + x0;
                r[i] = blend_group<SRC>(a, d[i], sw[i], make_uint2(0, 0), (uint32_t)p, 0xFFFFFFFFu);
                if(NT) nt_store(r[i], reinterpret_cast<uint4 *>(dst + p));
                else *reinterpret_cast<uint4 *>(dst + p) = r[i];
            }
        }
        const bool r90 = a.rot == 1;
#pragma unroll
        for(int j = 0; j < 8; j++) {
            uint32_t o[4];
#pragma unroll
            for(int m = 0; m < 4; m++) {
                auto word = [j](const uint4 & v) { return j < 2 ? v.x : (j < 4 ? v.y : (j < 6 ? v.z : v.w)); };
                const uint32_t lo = r90 ? word(r[7 - 2 * m]) : word(r[2 * m]);
                const uint32_t hi = r90 ? word(r[6 - 2 * m]) : word(r[2 * m + 1]);
                o[m] = (j & 1) ? __builtin_amdgcn_perm(hi, lo, 0x07060302u) : __builtin_amdgcn_perm(hi, lo, 0x05040100u);
            }
            const int xl = bx * 8 + j;
            lds[wave][xl * LY + (by ^ (bx & (LY - 1)))] = make_uint4(o[0], o[1], o[2], o[3]);
        }
    }
    __syncthreads();
    if(live && a.out) {
        uint16_t * out = a.out + (long long)f * a.out_fs;
#pragma unroll
        for(int k = 0; k < 8; k++) {
            const int q = k * 64 + lane;
            const int xl = q / LY, c = q % LY;
            const int X = tx * TX + xl;
            const int yb = ty * TY + c * 8;
            if(X < W && yb < H) {
                uint4 v = lds[wave][xl * LY + (c ^ ((xl >> 3) & (LY - 1)))];
                size_t o = a.rot == 1 ? (size_t)X * H + (size_t)(H - 8 - yb) : (size_t)(W - 1 - X) * H + (size_t)yb;
                if(NT) nt_store(v, reinterpret_cast<uint4 *>(out + o));
                else *reinterpret_cast<uint4 *>(out + o) = v;
            }
        }
    }
}

template <int SRC, int LX, bool NT, int REMAP, class R>
void run_v1(const char * name, BlendArgs a, int W, int H, int n_frames, size_t px, Timer & T, R & report, double bpp)
{
    constexpr int LY = 64 / LX;
    const uint32_t tiles_x = (W + 8 * LX - 1) / (8 * LX), tiles_y = (H + 8 * LY - 1) / (8 * LY), tpf = tiles_x * tiles_y;
    const uint32_t n_tiles = tpf * n_frames, nb = (n_tiles + 3) / 4;
    float ms = T.run([&] { hipLaunchKernelGGL((k_tile_v1<SRC, LX, NT, REMAP>), dim3(nb), dim3(256), 0, 0, a, W, H, tiles_x, tpf, n_tiles, nb); });
    report(name, px * bpp, ms);
}

// linear blend with XCD remap
template <int SRC, int REMAP, int UNR>
__global__ __launch_bounds__(256) void k_lin_v(BlendArgs a, uint32_t gpf, uint32_t cpf, uint32_t n_blocks)
{
    uint32_t blk = blockIdx.x;
    if(REMAP) {
        const uint32_t q = n_blocks / 8, r = n_blocks % 8, x = blk % 8;
        blk = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + blk / 8;
    }
    const uint32_t f = blk / cpf;
    const uint32_t c = blk - f * cpf;
    uint16_t * dst = a.dst + (long long)f * a.dst_fs;
    const long long src_px0 = (long long)f * a.src_fs;
    const uint32_t g0 = c * (256 * UNR) + threadIdx.x;
    uint4 d[UNR];
    uint32_t sw[UNR][SrcWords<SRC>::N];
#pragma unroll
    for(int u = 0; u < UNR; u++) {
        const uint32_t g = g0 + u * 256;
        if(g < gpf) {
            d[u] = *reinterpret_cast<const uint4 *>(dst + (size_t)g * 8);
            load_src<SRC>(a.src, src_px0 + (long long)g * 8, sw[u]);
        }
    }
#pragma unroll
    for(int u = 0; u < UNR; u++) {
        const uint32_t g = g0 + u * 256;
        if(g < gpf) {
            uint4 r = blend_group<SRC>(a, d[u], sw[u], make_uint2(0, 0), g * 8, 0xFFFFFFFFu);
            *reinterpret_cast<uint4 *>(dst + (size_t)g * 8) = r;
            if(a.out) *reinterpret_cast<uint4 *>(a.out + (long long)f * a.out_fs + (size_t)g * 8) = r;
        }
    }
}
template <int REMAP, int UNR, class R>
void run_lin(const char * name, BlendArgs a, int W, int H, int n_frames, size_t px, Timer & T, R & report, double bpp)
{
    const uint32_t gpf = (uint32_t)((size_t)W * H / 8), cpf = (gpf + 256 * UNR - 1) / (256 * UNR);
    const uint32_t nb = cpf * n_frames;
    float ms = T.run([&] { hipLaunchKernelGGL((k_lin_v<SRC_ARGB8888, REMAP, UNR>), dim3(nb), dim3(256), 0, 0, a, gpf, cpf, nb); });
    report(name, px * bpp, ms);
}


// ---- V2: waves-per-block and two-phase row loading ---------------------------------------------------------------
template <int SRC, int WPB, int PHASES>
__global__ __launch_bounds__(64 * WPB) void k_tile_v2(BlendArgs a, int W, int H, uint32_t tiles_x, uint32_t tpf, uint32_t n_tiles,
                                                      uint32_t n_blocks)
{
    __shared__ uint4 lds[WPB][512];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const uint32_t blk = xcd_remap(blockIdx.x, n_blocks);
    const uint32_t wt = __builtin_amdgcn_readfirstlane(blk * WPB + wave);
    const bool live = wt < n_tiles;
    const uint32_t f = live ? wt / tpf : 0;
    const uint32_t t = live ? wt - f * tpf : 0;
    const uint32_t ty = t / tiles_x, tx = t - ty * tiles_x;
    const int bx = lane & 7, by = lane >> 3;
    const int x0 = tx * 64 + bx * 8, y0 = ty * 64 + by * 8;
    const bool inside = live && x0 < W && y0 < H;
    uint4 r[8];
    if(inside) {
        uint16_t * dst = a.dst + (long long)f * a.dst_fs;
        const long long src_px0 = (long long)f * a.src_fs;
        constexpr int RPP = 8 / PHASES;
#pragma unroll
        for(int ph = 0; ph < PHASES; ph++) {
            uint4 d[RPP];
            uint32_t sw[RPP][SrcWords<SRC>::N];
#pragma unroll
            for(int i = 0; i < RPP; i++) {
                const size_t p = (size_t)(y0 + ph * RPP + i) * W + x0;
                d[i] = *reinterpret_cast<const uint4 *>(dst + p);
                load_src<SRC>(a.src, src_px0 + (long long)p, sw[i]);
            }
#pragma unroll
            for(int i = 0; i < RPP; i++) {
                const size_t p = (size_t)(y0 + ph * RPP + i) * W + x0;
                r[ph * RPP + i] = blend_group<SRC>(a, d[i], sw[i], make_uint2(0, 0), (uint32_t)p, 0xFFFFFFFFu);
                *reinterpret_cast<uint4 *>(dst + p) = r[ph * RPP + i];
            }
        }
#pragma unroll
        for(int j = 0; j < 8; j++) {
            uint32_t o[4];
#pragma unroll
            for(int m = 0; m < 4; m++) {
                auto word = [j](const uint4 & v) { return j < 2 ? v.x : (j < 4 ? v.y : (j < 6 ? v.z : v.w)); };
                const uint32_t lo = word(r[7 - 2 * m]);
                const uint32_t hi = word(r[6 - 2 * m]);
                o[m] = (j & 1) ? __builtin_amdgcn_perm(hi, lo, 0x07060302u) : __builtin_amdgcn_perm(hi, lo, 0x05040100u);
            }
            const int xl = bx * 8 + j;
            lds[wave][xl * 8 + (by ^ bx)] = make_uint4(o[0], o[1], o[2], o[3]);
        }
    }
    __syncthreads();
    if(live && a.out) {
        uint16_t * out = a.out + (long long)f * a.out_fs;
        const int rr = lane >> 3, c = lane & 7;
        const int yb = ty * 64 + c * 8;
#pragma unroll
        for(int k = 0; k < 8; k++) {
            const int xl = k * 8 + rr;
            const int X = tx * 64 + xl;
            if(X < W && yb < H) {
                uint4 v = lds[wave][xl * 8 + (c ^ k)];
                *reinterpret_cast<uint4 *>(out + (size_t)X * H + (size_t)(H - 8 - yb)) = v;
            }
        }
    }
}
template <int WPB, int PHASES, class R>
void run_v2(const char * name, BlendArgs a, int W, int H, int n_frames, size_t px, Timer & T, R & report)
{
    const uint32_t tiles_x = (W + 63) / 64, tiles_y = (H + 63) / 64, tpf = tiles_x * tiles_y;
    const uint32_t n_tiles = tpf * n_frames, nb = (n_tiles + WPB - 1) / WPB;
    float ms = T.run([&] { hipLaunchKernelGGL((k_tile_v2<SRC_ARGB8888, WPB, PHASES>), dim3(nb), dim3(64 * WPB), 0, 0, a, W, H, tiles_x, tpf, n_tiles, nb); });
    report(name, px * 10.0, ms);
}

// ---- V3: a 4-wave block owns a 128x128 tile; 32 KB LDS image [128 out rows][128 px]; 256-byte output runs --------
template <int SRC>
__global__ __launch_bounds__(256) void k_tile_v3(BlendArgs a, int W, int H, uint32_t tiles_x, uint32_t tpf, uint32_t n_tiles,
                                                 uint32_t n_blocks)
{
    __shared__ uint4 lds[128 * 16];  // out row xl (0..127) x 16 chunks of 8 px
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const uint32_t t_all = xcd_remap(blockIdx.x, n_blocks);
    const uint32_t f = t_all / tpf;
    const uint32_t t = t_all - f * tpf;
    const uint32_t ty = t / tiles_x, tx = t - ty * tiles_x;
    const int qx = wave & 1, qy = wave >> 1;  // quadrant
    const int bx = lane & 7, by = lane >> 3;
    const int xl0 = qx * 64 + bx * 8, yl0 = qy * 64 + by * 8;  // tile-local
    const int x0 = tx * 128 + xl0, y0 = ty * 128 + yl0;
    const bool inside = x0 < W && y0 < H;
    if(inside) {
        uint16_t * dst = a.dst + (long long)f * a.dst_fs;
        const long long src_px0 = (long long)f * a.src_fs;
        uint4 d[8], r[8];
        uint32_t sw[8][SrcWords<SRC>::N];
#pragma unroll
        for(int i = 0; i < 8; i++) {
            const size_t p = (size_t)(y0 + i) * W + x0;
            d[i] = *reinterpret_cast<const uint4 *>(dst + p);
            load_src<SRC>(a.src, src_px0 + (long long)p, sw[i]);
        }
#pragma unroll
        for(int i = 0; i < 8; i++) {
            const size_t p = (size_t)(y0 + i) * W + x0;
            r[i] = blend_group<SRC>(a, d[i], sw[i], make_uint2(0, 0), (uint32_t)p, 0xFFFFFFFFu);
            *reinterpret_cast<uint4 *>(dst + p) = r[i];
        }
#pragma unroll
        for(int j = 0; j < 8; j++) {
            uint32_t o[4];
#pragma unroll
            for(int m = 0; m < 4; m++) {
                auto word = [j](const uint4 & v) { return j < 2 ? v.x : (j < 4 ? v.y : (j < 6 ? v.z : v.w)); };
                const uint32_t lo = word(r[7 - 2 * m]);
                const uint32_t hi = word(r[6 - 2 * m]);
                o[m] = (j & 1) ? __builtin_amdgcn_perm(hi, lo, 0x07060302u) : __builtin_amdgcn_perm(hi, lo, 0x05040100u);
            }
            const int xl = xl0 + j;
            // chunk index along the OUTPUT row: rot90 reverses y, chunk c' = 15 - (yl0/8)
            const int c = 15 - (yl0 >> 3);
            lds[xl * 16 + (c ^ ((xl >> 3) & 15))] = make_uint4(o[0], o[1], o[2], o[3]);
        }
    }
    __syncthreads();
    if(a.out) {
        uint16_t * out = a.out + (long long)f * a.out_fs;
#pragma unroll
        for(int k = 0; k < 8; k++) {
            const int q = k * 256 + threadIdx.x;   // 2048 (row, chunk) pairs
            const int xl = q >> 4, c = q & 15;
            const int X = tx * 128 + xl;
            const int yb = ty * 128 + (15 - c) * 8;  // source rows of this chunk
            if(X < W && yb < H) {
                uint4 v = lds[xl * 16 + (c ^ ((xl >> 3) & 15))];
                *reinterpret_cast<uint4 *>(out + (size_t)X * H + (size_t)(H - 8 - yb)) = v;
            }
        }
    }
}
template <class R>
void run_v3(const char * name, BlendArgs a, int W, int H, int n_frames, size_t px, Timer & T, R & report)
{
    const uint32_t tiles_x = (W + 127) / 128, tiles_y = (H + 127) / 128, tpf = tiles_x * tiles_y;
    const uint32_t nb = tpf * n_frames;
    float ms = T.run([&] { hipLaunchKernelGGL((k_tile_v3<SRC_ARGB8888>), dim3(nb), dim3(256), 0, 0, a, W, H, tiles_x, tpf, nb, nb); });
    report(name, px * 10.0, ms);
}


// ---- V4: selective non-temporal accesses on the production tile layout ------------------------------------------
// NTMASK bit0: src loads, bit1: dst loads, bit2: dst stores, bit3: out stores
template <int SRC, int NTMASK>
__global__ __launch_bounds__(256) void k_tile_v4(BlendArgs a, int W, int H, uint32_t tiles_x, uint32_t tpf, uint32_t n_tiles,
                                                 uint32_t n_blocks)
{
    __shared__ uint4 lds[4][512];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const uint32_t blk = xcd_remap(blockIdx.x, n_blocks);
    const uint32_t wt = __builtin_amdgcn_readfirstlane(blk * 4 + wave);
    const bool live = wt < n_tiles;
    const uint32_t f = live ? wt / tpf : 0;
    const uint32_t t = live ? wt - f * tpf : 0;
    const uint32_t ty = t / tiles_x, tx = t - ty * tiles_x;
    const int bx = lane & 7, by = lane >> 3;
    const int x0 = tx * 64 + bx * 8, y0 = ty * 64 + by * 8;
    const bool inside = live && x0 < W && y0 < H;
    uint4 r[8];
    if(inside) {
        uint16_t * dst = a.dst + (long long)f * a.dst_fs;
        const long long src_px0 = (long long)f * a.src_fs;
        uint4 d[8];
        uint32_t sw[8][8];
#pragma unroll
        for(int i = 0; i < 8; i++) {
            const size_t p = (size_t)(y0 + i) * W + x0;
            d[i] = (NTMASK & 2) ? nt_load(reinterpret_cast<const uint4 *>(dst + p)) : *reinterpret_cast<const uint4 *>(dst + p);
            const uint4 * pp = reinterpret_cast<const uint4 *>(reinterpret_cast<const uint32_t *>(a.src) + src_px0 + p);
            uint4 v0 = (NTMASK & 1) ? nt_load(pp) : pp[0], v1 = (NTMASK & 1) ? nt_load(pp + 1) : pp[1];
            sw[i][0] = v0.x; sw[i][1] = v0.y; sw[i][2] = v0.z; sw[i][3] = v0.w;
            sw[i][4] = v1.x; sw[i][5] = v1.y; sw[i][6] = v1.z; sw[i][7] = v1.w;
        }
#pragma unroll
        for(int i = 0; i < 8; i++) {
            const size_t p = (size_t)(y0 + i) * W + x0;
            r[i] = blend_group<SRC>(a, d[i], sw[i], make_uint2(0, 0), (uint32_t)p, 0xFFFFFFFFu);
            if(NTMASK & 4) nt_store(r[i], reinterpret_cast<uint4 *>(dst + p));
            else *reinterpret_cast<uint4 *>(dst + p) = r[i];
        }
#pragma unroll
        for(int j = 0; j < 8; j++) {
            uint32_t o[4];
#pragma unroll
            for(int m = 0; m < 4; m++) {
                auto word = [j](const uint4 & v) { return j < 2 ? v.x : (j < 4 ? v.y : (j < 6 ? v.z : v.w)); };
                const uint32_t lo = word(r[7 - 2 * m]);
                const uint32_t hi = word(r[6 - 2 * m]);
                o[m] = (j & 1) ? __builtin_amdgcn_perm(hi, lo, 0x07060302u) : __builtin_amdgcn_perm(hi, lo, 0x05040100u);
            }
            const int xl = bx * 8 + j;
            lds[wave][xl * 8 + (by ^ bx)] = make_uint4(o[0], o[1], o[2], o[3]);
        }
    }
    __syncthreads();
    if(live && a.out) {
        uint16_t * out = a.out + (long long)f * a.out_fs;
        const int rr = lane >> 3, c = lane & 7;
        const int yb = ty * 64 + c * 8;
#pragma unroll
        for(int k = 0; k < 8; k++) {
            const int xl = k * 8 + rr;
            const int X = tx * 64 + xl;
            if(X < W && yb < H) {
                uint4 v = lds[wave][xl * 8 + (c ^ k)];
                uint4 * op = reinterpret_cast<uint4 *>(out + (size_t)X * H + (size_t)(H - 8 - yb));
                if(NTMASK & 8) nt_store(v, op);
                else *op = v;
            }
        }
    }
}
template <int NTMASK, class R>
void run_v4(const char * name, BlendArgs a, int W, int H, int n_frames, size_t px, Timer & T, R & report)
{
    const uint32_t tiles_x = (W + 63) / 64, tiles_y = (H + 63) / 64, tpf = tiles_x * tiles_y;
    const uint32_t n_tiles = tpf * n_frames, nb = (n_tiles + 3) / 4;
    float ms = T.run([&] { hipLaunchKernelGGL((k_tile_v4<SRC_ARGB8888, NTMASK>), dim3(nb), dim3(256), 0, 0, a, W, H, tiles_x, tpf, n_tiles, nb); });
    report(name, px * 10.0, ms);
}

// ---- V5: linear blend, not in place (reads dst+src, writes only out) ---------------------------------------------
template <int MODE>  // 0: in place (8 B/px), 1: result to out only (8 B/px), 2: in place + out (10 B/px), 3: read-only (6 B/px)
__global__ __launch_bounds__(256) void k_lin_v5(BlendArgs a, uint32_t gpf, uint32_t cpf, uint32_t n_blocks)
{
    const uint32_t blk = xcd_remap(blockIdx.x, n_blocks);
    const uint32_t f = blk / cpf;
    const uint32_t c = blk - f * cpf;
    uint16_t * dst = a.dst + (long long)f * a.dst_fs;
    const long long src_px0 = (long long)f * a.src_fs;
    const uint32_t g0 = c * 512 + threadIdx.x;
    uint4 d[2];
    uint32_t sw[2][8];
#pragma unroll
    for(int u = 0; u < 2; u++) {
        const uint32_t g = g0 + u * 256;
        if(g < gpf) {
            d[u] = *reinterpret_cast<const uint4 *>(dst + (size_t)g * 8);
            load_src<SRC_ARGB8888>(a.src, src_px0 + (long long)g * 8, sw[u]);
        }
    }
#pragma unroll
    for(int u = 0; u < 2; u++) {
        const uint32_t g = g0 + u * 256;
        if(g < gpf) {
            uint4 r = blend_group<SRC_ARGB8888>(a, d[u], sw[u], make_uint2(0, 0), g * 8, 0xFFFFFFFFu);
            if(MODE == 0 || MODE == 2) *reinterpret_cast<uint4 *>(dst + (size_t)g * 8) = r;
            if(MODE == 1 || MODE == 2) *reinterpret_cast<uint4 *>(a.out + (long long)f * a.out_fs + (size_t)g * 8) = r;
            if(MODE == 3 && r.x == 0x12345678u && r.y == 0x3456789au && r.z == 0xdeadbeefu) *reinterpret_cast<uint4 *>(dst + (size_t)g * 8) = r;
        }
    }
}
template <int MODE, class R>
void run_v5(const char * name, BlendArgs a, int W, int H, int n_frames, size_t px, Timer & T, R & report, double bpp)
{
    const uint32_t gpf = (uint32_t)((size_t)W * H / 8), cpf = (gpf + 511) / 512;
    const uint32_t nb = cpf * n_frames;
    float ms = T.run([&] { hipLaunchKernelGGL((k_lin_v5<MODE>), dim3(nb), dim3(256), 0, 0, a, gpf, cpf, nb); });
    report(name, px * bpp, ms);
}


// ---- V6: remap granularity: XCD x takes groups of G consecutive frames round-robin (G*8 frames per round) ------------
template <int SRC, int NTOUT>
__global__ __launch_bounds__(256) void k_tile_v6(BlendArgs a, int W, int H, uint32_t tiles_x, uint32_t tpf, uint32_t n_tiles,
                                                 uint32_t n_blocks, uint32_t bpg /*blocks per group of frames*/)
{
    __shared__ uint4 lds[4][512];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    uint32_t blk;
    {
        const uint32_t x = blockIdx.x & 7u, i = blockIdx.x >> 3;
        const uint32_t grp = i / bpg, within = i - grp * bpg;
        blk = (grp * 8 + x) * bpg + within;
    }
    const uint32_t wt = __builtin_amdgcn_readfirstlane(blk * 4 + wave);
    const bool live = wt < n_tiles;
    const uint32_t f = live ? wt / tpf : 0;
    const uint32_t t = live ? wt - f * tpf : 0;
    const uint32_t ty = t / tiles_x, tx = t - ty * tiles_x;
    const int bx = lane & 7, by = lane >> 3;
    const int x0 = tx * 64 + bx * 8, y0 = ty * 64 + by * 8;
    const bool inside = live && x0 < W && y0 < H;
    uint4 r[8];
    if(inside) {
        uint16_t * dst = a.dst + (long long)f * a.dst_fs;
        const long long src_px0 = (long long)f * a.src_fs;
        uint4 d[8];
        uint32_t sw[8][8];
#pragma unroll
        for(int i = 0; i < 8; i++) {
            const size_t p = (size_t)(y0 + i) * W + x0;
            d[i] = *reinterpret_cast<const uint4 *>(dst + p);
            load_src<SRC>(a.src, src_px0 + (long long)p, sw[i]);
        }
#pragma unroll
        for(int i = 0; i < 8; i++) {
            const size_t p = (size_t)(y0 + i) * W + x0;
            r[i] = blend_group<SRC>(a, d[i], sw[i], make_uint2(0, 0), (uint32_t)p, 0xFFFFFFFFu);
            *reinterpret_cast<uint4 *>(dst + p) = r[i];
        }
#pragma unroll
        for(int j = 0; j < 8; j++) {
            uint32_t o[4];
#pragma unroll
            for(int m = 0; m < 4; m++) {
                auto word = [j](const uint4 & v) { return j < 2 ? v.x : (j < 4 ? v.y : (j < 6 ? v.z : v.w)); };
                const uint32_t lo = word(r[7 - 2 * m]);
                const uint32_t hi = word(r[6 - 2 * m]);
                o[m] = (j & 1) ? __builtin_amdgcn_perm(hi, lo, 0x07060302u) : __builtin_amdgcn_perm(hi, lo, 0x05040100u);
            }
            const int xl = bx * 8 + j;
            lds[wave][xl * 8 + (by ^ bx)] = make_uint4(o[0], o[1], o[2], o[3]);
        }
    }
    __syncthreads();
    if(live && a.out) {
        uint16_t * out = a.out + (long long)f * a.out_fs;
        const int rr = lane >> 3, c = lane & 7;
        const int yb = ty * 64 + c * 8;
#pragma unroll
        for(int k = 0; k < 8; k++) {
            const int xl = k * 8 + rr;
            const int X = tx * 64 + xl;
            if(X < W && yb < H) {
                uint4 v = lds[wave][xl * 8 + (c ^ k)];
                uint4 * op = reinterpret_cast<uint4 *>(out + (size_t)X * H + (size_t)(H - 8 - yb));
                if(NTOUT) nt_store(v, op);
                else *op = v;
            }
        }
    }
}
template <int NTOUT, class R>
void run_v6(const char * name, int frames_per_group, BlendArgs a, int W, int H, int n_frames, size_t px, Timer & T, R & report)
{
    const uint32_t tiles_x = (W + 63) / 64, tiles_y = (H + 63) / 64, tpf = tiles_x * tiles_y;
    const uint32_t n_tiles = tpf * n_frames, nb = (n_tiles + 3) / 4;
    const uint32_t bpg = (tpf / 4) * frames_per_group;   // needs tpf % 4 == 0 and n_frames % (8*frames_per_group) == 0
    float ms = T.run([&] { hipLaunchKernelGGL((k_tile_v6<SRC_ARGB8888, NTOUT>), dim3(nb), dim3(256), 0, 0, a, W, H, tiles_x, tpf, n_tiles, nb, bpg); });
    report(name, px * 10.0, ms);
}

template <class R>
void run_variants(BlendArgs a, int W, int H, int n_frames, size_t px, Timer & T, R & report)
{
    run_v4<0>("v4 fused contiguous remap", a, W, H, n_frames, px, T, report);
    run_v4<8>("v4 fused contiguous remap, nt out", a, W, H, n_frames, px, T, report);
    run_v6<0>("v6 remap 1 frame per xcd turn", 1, a, W, H, n_frames, px, T, report);
    run_v6<0>("v6 remap 4 frames per xcd turn", 4, a, W, H, n_frames, px, T, report);
    run_v6<0>("v6 remap 16 frames per xcd turn", 16, a, W, H, n_frames, px, T, report);
    run_v6<0>("v6 remap 64 frames per xcd turn", 64, a, W, H, n_frames, px, T, report);
    run_v6<1>("v6 remap 16 frames per xcd turn, nt out", 16, a, W, H, n_frames, px, T, report);
    run_v6<1>("v6 remap 64 frames per xcd turn, nt out", 64, a, W, H, n_frames, px, T, report);
}
