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

template <class R>
void run_variants(BlendArgs a, int W, int H, int n_frames, size_t px, Timer & T, R & report)
{
    run_v1<SRC_ARGB8888, 8, false, 1>("v1 fused 64x64 xcd-remap", a, W, H, n_frames, px, T, report, 10.0);
    run_v1<SRC_ARGB8888, 4, false, 0>("v1 fused 32x128", a, W, H, n_frames, px, T, report, 10.0);
    run_v1<SRC_ARGB8888, 4, false, 1>("v1 fused 32x128 xcd-remap", a, W, H, n_frames, px, T, report, 10.0);
    run_v1<SRC_ARGB8888, 2, false, 1>("v1 fused 16x256 xcd-remap", a, W, H, n_frames, px, T, report, 10.0);
    run_v1<SRC_COPY, 8, false, 1>("v1 rotate 64x64 xcd-remap", a, W, H, n_frames, px, T, report, 4.0);
    run_v1<SRC_COPY, 4, false, 0>("v1 rotate 32x128", a, W, H, n_frames, px, T, report, 4.0);
    run_v1<SRC_COPY, 4, false, 1>("v1 rotate 32x128 xcd-remap", a, W, H, n_frames, px, T, report, 4.0);
    run_v1<SRC_COPY, 2, false, 1>("v1 rotate 16x256 xcd-remap", a, W, H, n_frames, px, T, report, 4.0);
    BlendArgs b = a;
    b.out = nullptr;
    run_lin<0, 4>("lin blend unroll4", b, W, H, n_frames, px, T, report, 8.0);
    run_lin<1, 4>("lin blend unroll4 xcd-remap", b, W, H, n_frames, px, T, report, 8.0);
    run_lin<1, 2>("lin blend unroll2 xcd-remap", b, W, H, n_frames, px, T, report, 8.0);
    run_lin<1, 8>("lin blend unroll8 xcd-remap", b, W, H, n_frames, px, T, report, 8.0);
    run_lin<1, 4>("lin blend+copy unroll4 xcd-remap", a, W, H, n_frames, px, T, report, 10.0);
}
