// ubench5.hip -- what bounds the ARGB8565-destination masked fill (development tool, not part of the product).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/ubench5 tools/ubench5.hip && ./tools/ubench5
// One persistent kernel shape (the 3 x 1-KB exchange of k_blend_argb_linear, workgroups walking chunks with a stride of
// the grid) with the per-pixel work exchanged: COMPUTE 0 = none (the memory pattern alone), 2 = the arithmetic of
// k_blend_argb_linear; beside plain 16-byte streams with the same byte mix (3 B/px read and written, 1 B/px read).
// Measured (4096 x 480 x 320, % of 8 TB/s): plain streams 67.9 out of place / 66.7 in place; the pattern without
// arithmetic 61.6-64.8; with it 60.0-60.7 -- the launch sits within 5 points of what its access pattern reaches, and
// the arithmetic costs little of that.  (A variant with lv_color_mix_with_alpha's opacity arithmetic tabulated in
// 128 KB of LDS -- 21 instead of 50 instructions per pixel -- reached 52.8 here: one 8-wave workgroup per CU and a
// random 2-byte LDS read per pixel are worse than the instructions they replace.)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../vvc_amd/csrc/lvgpu_kernels.hpp"
using namespace lvgpu;
#define CK(x) do { hipError_t e = (x); if(e != hipSuccess) { fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e)); exit(1); } } while(0)

template <int WAVES, bool PREFETCH, int COMPUTE>
__global__ __launch_bounds__(64 * WAVES) void k_var(BlendArgs a, uint32_t gpf, uint32_t cpf, uint32_t n_chunks)
{
    extern __shared__ uint4 lds_dyn[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint4 * xch = lds_dyn + wave * 192;
    const uint32_t n_words = gpf * 3u;
    struct In { uint4 v[3]; uint2 mk[2]; uint32_t gw, f; };
    auto issue = [&](uint32_t ch, In & in) {
        const uint32_t f = ch / cpf, c = ch - f * cpf;
        in.f = f;
        in.gw = c * (WAVES * 64) + (uint32_t)wave * 64;
        in.mk[0] = in.mk[1] = make_uint2(0, 0);
        if(in.gw >= gpf) return;
        const uint4 * dst16 = reinterpret_cast<const uint4 *>(reinterpret_cast<const uint8_t *>(a.dst) + (long long)f * a.dst_fs * 3);
#pragma unroll
        for(int i = 0; i < 3; i++) {
            const uint32_t word = in.gw * 3u + (uint32_t)(i * 64 + lane);
            in.v[i] = word < n_words ? dst16[word] : make_uint4(0, 0, 0, 0);
        }
        const uint32_t g = in.gw + (uint32_t)lane;
        if(g < gpf && a.mask) {
            const uint4 m = *reinterpret_cast<const uint4 *>(a.mask + (long long)f * a.mask_fs + (size_t)g * 16);
            in.mk[0] = make_uint2(m.x, m.y);
            in.mk[1] = make_uint2(m.z, m.w);
        }
    };
    auto blend = [&](In & in) {
        if(in.gw >= gpf) return;
        uint4 * dst16 = reinterpret_cast<uint4 *>(reinterpret_cast<uint8_t *>(a.dst) + (long long)in.f * a.dst_fs * 3);
#pragma unroll
        for(int i = 0; i < 3; i++) xch[i * 64 + lane] = in.v[i];
        wave_lds_fence();
        const uint4 w0 = xch[3 * lane], w1 = xch[3 * lane + 1], w2 = xch[3 * lane + 2];
        uint32_t dw[2][6];
        dw[0][0] = w0.x; dw[0][1] = w0.y; dw[0][2] = w0.z; dw[0][3] = w0.w; dw[0][4] = w1.x; dw[0][5] = w1.y;
        dw[1][0] = w1.z; dw[1][1] = w1.w; dw[1][2] = w2.x; dw[1][3] = w2.y; dw[1][4] = w2.z; dw[1][5] = w2.w;
        wave_lds_fence();
        if(in.gw + (uint32_t)lane < gpf) {
#pragma unroll
            for(int h = 0; h < 2; h++) {
                uint32_t sw[1] = {0};
                if(COMPUTE == 0) { dw[h][0] ^= in.mk[h].x; dw[h][3] ^= in.mk[h].y; }
                else argb_group_normal<SRC_FILL, AM_MASK>(a, dw[h], sw, in.mk[h], a.color);
            }
        }
        xch[3 * lane] = make_uint4(dw[0][0], dw[0][1], dw[0][2], dw[0][3]);
        xch[3 * lane + 1] = make_uint4(dw[0][4], dw[0][5], dw[1][0], dw[1][1]);
        xch[3 * lane + 2] = make_uint4(dw[1][2], dw[1][3], dw[1][4], dw[1][5]);
        wave_lds_fence();
#pragma unroll
        for(int i = 0; i < 3; i++) {
            const uint32_t word = in.gw * 3u + (uint32_t)(i * 64 + lane);
            if(word < n_words) dst16[word] = xch[i * 64 + lane];
        }
        wave_lds_fence();
    };
    if(PREFETCH) {
        uint32_t ch = blockIdx.x;
        In cur, nxt;
        issue(ch, cur);
        while(true) {
            const uint32_t next = ch + gridDim.x;
            const bool more = next < n_chunks;
            if(more) issue(next, nxt);
            blend(cur);
            if(!more) break;
            cur = nxt;
            ch = next;
        }
    }
    else {
        for(uint32_t ch = blockIdx.x; ch < n_chunks; ch += gridDim.x) {
            In cur;
            issue(ch, cur);
            blend(cur);
        }
    }
}

// plain streams with the same byte mix: 3 B/px read + written, 1 B/px read, 16 bytes per lane and access
__global__ __launch_bounds__(256) void k_streams(const uint4 * d_in, uint4 * d_out, const uint4 * m, size_t n16)
{
    // per 16 pixels: three words of dst, one of mask
    const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
    if(g >= n16) return;
    const uint4 mm = m[g];
    const size_t wave0 = (g & ~(size_t)63) * 3, lane = g & 63;
#pragma unroll
    for(int i = 0; i < 3; i++) {
        uint4 v = d_in[wave0 + i * 64 + lane];
        v.x ^= mm.x;
        d_out[wave0 + i * 64 + lane] = v;
    }
}

int main(int argc, char ** argv)
{
    const int n_frames = argc > 1 ? atoi(argv[1]) : 4096;
    const int W = 480, H = 320;
    const size_t px = (size_t)W * H * n_frames;
    uint8_t *dst, *mask;
    CK(hipMalloc(&dst, px * 3));
    CK(hipMalloc(&mask, px));
    hipLaunchKernelGGL(k_synth, dim3(4096), dim3(256), 0, 0, dst, 0, 1ull, 0ull, px * 3 / 2);
    hipLaunchKernelGGL(k_synth, dim3(4096), dim3(256), 0, 0, mask, 0, 2ull, 0ull, px / 2);
    CK(hipDeviceSynchronize());
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    BlendArgs a;
    memset(&a, 0, sizeof(a));
    a.dst = (uint16_t *)dst; a.dst_fs = (long long)W * H; a.dst_stride = W;
    a.mask = mask; a.mask_fs = (long long)W * H; a.mask_stride = W;
    a.w = W; a.h = H; a.n_frames = n_frames; a.color = 0x1234; a.opa = 255; a.raw_opa = 255; a.has_mask = 1; a.swap = 1;
    const uint32_t gpf = W * H / 16;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto timeit = [&](const char * name, auto launch) {
        for(int i = 0; i < 3; i++) launch();
        CK(hipDeviceSynchronize());
        std::vector<float> ts;
        for(int i = 0; i < 10; i++) {
            CK(hipEventRecord(e0));
            launch();
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            ts.push_back(ms);
        }
        CK(hipGetLastError());
        std::sort(ts.begin(), ts.end());
        const float ms = ts[ts.size() / 2];
        printf("%-58s %7.4f ms %6.1f %% of 8 TB/s\n", name, ms, px * 7.0 / ms / 1e6 / 80.0);
        fflush(stdout);
    };
#define VAR(WAVES, PRE, COMP, GRIDMUL)                                                                                        \
    {                                                                                                                         \
        const uint32_t cpf = (gpf + WAVES * 64 - 1) / (WAVES * 64);                                                           \
        const uint32_t n_chunks = cpf * n_frames;                                                                             \
        const uint32_t lds = WAVES * 3072u;                                                                                   \
        auto k = k_var<WAVES, PRE, COMP>;                                                                                     \
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));     \
        char name[128];                                                                                                       \
        snprintf(name, sizeof name, "waves %d prefetch %d compute %d grid %dxCUs", WAVES, (int)PRE, COMP, GRIDMUL); \
        timeit(name, [&] { hipLaunchKernelGGL(k, dim3(cus * GRIDMUL), dim3(64 * WAVES), lds, 0, a, gpf, cpf, n_chunks); }); \
    }
    {
        const size_t n16 = px / 16;
        uint8_t * out;
        CK(hipMalloc(&out, px * 3));
        timeit("plain streams, same byte mix, out of place", [&] { hipLaunchKernelGGL(k_streams, dim3((n16 + 255) / 256), dim3(256), 0, 0, (const uint4 *)dst, (uint4 *)out, (const uint4 *)mask, n16); });
        timeit("plain streams, same byte mix, in place", [&] { hipLaunchKernelGGL(k_streams, dim3((n16 + 255) / 256), dim3(256), 0, 0, (const uint4 *)dst, (uint4 *)dst, (const uint4 *)mask, n16); });
        CK(hipFree(out));
    }
    VAR(8, true, 0, 1)
    VAR(8, true, 0, 2)
    VAR(8, true, 0, 4)
    VAR(8, false, 0, 4)
    VAR(4, true, 0, 8)
    VAR(4, false, 0, 8)
    VAR(4, false, 0, 16)
    VAR(8, true, 2, 1)
    VAR(8, true, 2, 2)
    VAR(8, true, 2, 4)
    VAR(4, true, 2, 8)
    VAR(4, false, 2, 8)
    VAR(4, false, 2, 16)
    VAR(16, true, 0, 1)
    VAR(16, true, 0, 2)
    VAR(16, true, 2, 1)
    VAR(16, true, 2, 2)
    return 0;
}
