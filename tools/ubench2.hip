// ubench2.hip -- development microbenchmark: does a persistent, software-pipelined sweep beat one short-lived workgroup
// per chunk for the 6 R : 4 W byte mix of the fused path?  (Not part of the product or the tests.)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/ubench2 tools/ubench2.hip && ./tools/ubench2
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#include "../vvc_amd/csrc/lvgpu_kernels.hpp"

using namespace lvgpu;

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e = (x);                                                            \
        if(e != hipSuccess) {                                                          \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e)); \
            exit(1);                                                                   \
        }                                                                              \
    } while(0)

struct Timer {
    hipEvent_t a, b;
    Timer()
    {
        CK(hipEventCreate(&a));
        CK(hipEventCreate(&b));
    }
    float run(const std::function<void()> & f, int warm = 3, int iters = 20)
    {
        for(int i = 0; i < warm; i++) f();
        CK(hipDeviceSynchronize());
        std::vector<float> ts;
        for(int i = 0; i < iters; i++) {
            CK(hipEventRecord(a));
            f();
            CK(hipEventRecord(b));
            CK(hipEventSynchronize(b));
            float ms;
            CK(hipEventElapsedTime(&ms, a, b));
            ts.push_back(ms);
        }
        std::sort(ts.begin(), ts.end());
        return ts[ts.size() / 2];
    }
};

// One contiguous run of n_groups 8-pixel groups: dst (RMW), ARGB8888 src, out (copy of the blended pixels).
// Persistent: XCD x (= block % 8) sweeps the x-th eighth of the run; its nb blocks take adjacent chunks each trip.
template <int U, bool PREFETCH>
__global__ __launch_bounds__(256) void k_lin_persist(BlendArgs a, uint32_t n_groups)
{
    constexpr uint32_t CH = 256 * U;
    const uint32_t n_chunks = (n_groups + CH - 1) / CH;
    const uint32_t x = blockIdx.x & 7u, bi = blockIdx.x >> 3, nb = gridDim.x >> 3;
    const uint32_t c_lo = (uint32_t)((uint64_t)n_chunks * x / 8), c_hi = (uint32_t)((uint64_t)n_chunks * (x + 1) / 8);
    uint4 d[U], dn[U];
    uint32_t sw[U][8], swn[U][8];
    auto fetch = [&](uint32_t c, uint4 * dd, uint32_t (*ss)[8]) {
#pragma unroll
        for(int u = 0; u < U; u++) {
            const uint32_t g = c * CH + u * 256 + threadIdx.x;
            if(g < n_groups) {
                dd[u] = *reinterpret_cast<const uint4 *>(a.dst + (size_t)g * 8);
                load_src<SRC_ARGB8888>(a.src, (long long)g * 8, ss[u]);
            }
        }
    };
    uint32_t c = c_lo + bi;
    if(c >= c_hi) return;
    fetch(c, d, sw);
    for(; c < c_hi; c += nb) {
        const uint32_t cn = c + nb;
        if(PREFETCH && cn < c_hi) fetch(cn, dn, swn);
#pragma unroll
        for(int u = 0; u < U; u++) {
            const uint32_t g = c * CH + u * 256 + threadIdx.x;
            if(g < n_groups) {
                uint4 r = blend_group_mix<SRC_ARGB8888, OM_MASK>(a, d[u], sw[u], make_uint2(0, 0));
                *reinterpret_cast<uint4 *>(a.dst + (size_t)g * 8) = r;
                store_streaming(a.out + (size_t)g * 8, r);
            }
        }
        if(PREFETCH) {
#pragma unroll
            for(int u = 0; u < U; u++) {
                d[u] = dn[u];
#pragma unroll
                for(int i = 0; i < 8; i++) sw[u][i] = swn[u][i];
            }
        }
        else if(cn < c_hi) fetch(cn, d, sw);
    }
}

int main(int argc, char ** argv)
{
    int n_frames = argc > 1 ? atoi(argv[1]) : 4096;
    const int W = 480, H = 320;
    const size_t px = (size_t)W * H * n_frames;
    uint16_t *dst, *out;
    uint32_t * src;
    CK(hipMalloc(&dst, px * 2));
    CK(hipMalloc(&out, px * 2));
    CK(hipMalloc(&src, px * 4));
    hipLaunchKernelGGL(k_synth, dim3(4096), dim3(256), 0, 0, dst, 0, 1ull, 0ull, px);
    hipLaunchKernelGGL(k_synth, dim3(4096), dim3(256), 0, 0, src, 3, 2ull, 0ull, px);
    CK(hipDeviceSynchronize());
    Timer T;
    auto report = [&](const char * name, double bytes, float ms) {
        printf("%-60s %8.4f ms  %8.1f GB/s  %5.1f%% of 8 TB/s\n", name, ms, bytes / ms / 1e6, bytes / ms / 1e6 / 80.0);
        fflush(stdout);
    };
    BlendArgs a;
    memset(&a, 0, sizeof(a));
    a.dst = dst;
    a.dst_fs = (long long)W * H;
    a.dst_stride = W;
    a.src = src;
    a.src_fs = (long long)W * H;
    a.src_stride = W;
    a.w = W;
    a.h = H;
    a.n_frames = n_frames;
    a.opa = 255;
    a.opa_mode = OM_MASK;
    a.swap = 1;
    a.out = out;
    a.out_fs = (long long)W * H;
    a.rot = 0;
    const uint32_t gpf = (uint32_t)((size_t)W * H / 8), cpf = (gpf + LIN_CHUNK - 1) / LIN_CHUNK;
    for(uint32_t pad : {0u, 32768u, 49152u}) {
        float ms = T.run([&] { hipLaunchKernelGGL(k_blend_linear<SRC_ARGB8888>, dim3(cpf * n_frames), dim3(256), pad, 0, a, gpf, cpf); });
        char nm[96];
        snprintf(nm, sizeof nm, "k_blend_linear + copy out (10 B/px), LDS pad %u", pad);
        report(nm, px * 10.0, ms);
    }
    const uint32_t n_groups = (uint32_t)(px / 8);
    char nm[96];
    for(int bpc : {1, 2, 3, 4, 6, 8}) {
        const uint32_t grid = 256 * bpc;
        float ms;
#define RUN(U, P)                                                                                                       \
    ms = T.run([&] { hipLaunchKernelGGL((k_lin_persist<U, P>), dim3(grid), dim3(256), 0, 0, a, n_groups); });          \
    snprintf(nm, sizeof nm, "persistent %d blocks/CU, %d groups/lane, prefetch %d", bpc, U, (int)P);                  \
    report(nm, px * 10.0, ms);
        RUN(1, false) RUN(1, true) RUN(2, false) RUN(2, true) RUN(4, false) RUN(4, true)
    }
    return 0;
}
