// ubench.hip -- kernel-variant microbenchmark (development tool, not part of the product or the tests).
// Builds against the production kernels and times candidate access patterns on one GPU with HIP events.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/ubench tools/ubench.hip && ./tools/ubench [frames]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
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

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 nt_load(const uint4 * p)
{
    u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void nt_store(uint4 v, uint4 * p)
{
    u32x4 t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<u32x4 *>(p));
}

// ---- plain streaming copy: the empirical HBM ceiling on this device -------------------------------------------
template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void k_copy(const uint4 * __restrict__ in, uint4 * __restrict__ out, size_t n)
{
    size_t base = (size_t)blockIdx.x * 256 * UNROLL + threadIdx.x;
    uint4 v[UNROLL];
#pragma unroll
    for(int u = 0; u < UNROLL; u++) {
        size_t i = base + (size_t)u * 256;
        if(i < n) v[u] = NT ? nt_load(in + i) : in[i];
    }
#pragma unroll
    for(int u = 0; u < UNROLL; u++) {
        size_t i = base + (size_t)u * 256;
        if(i < n) {
            if(NT) nt_store(v[u], out + i);
            else out[i] = v[u];
        }
    }
}

// read-only and write-only streams
template <int UNROLL>
__global__ __launch_bounds__(256) void k_read(const uint4 * __restrict__ in, uint4 * __restrict__ out, size_t n)
{
    size_t base = (size_t)blockIdx.x * 256 * UNROLL + threadIdx.x;
    uint4 acc = make_uint4(0, 0, 0, 0);
#pragma unroll
    for(int u = 0; u < UNROLL; u++) {
        size_t i = base + (size_t)u * 256;
        if(i < n) {
            uint4 v = in[i];
            acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
        }
    }
    if(acc.x == 0x12345678u && acc.y == 0x9abcdef0u) out[base] = acc;
}
template <int UNROLL>
__global__ __launch_bounds__(256) void k_write(uint4 * __restrict__ out, size_t n)
{
    size_t base = (size_t)blockIdx.x * 256 * UNROLL + threadIdx.x;
#pragma unroll
    for(int u = 0; u < UNROLL; u++) {
        size_t i = base + (size_t)u * 256;
        if(i < n) out[i] = make_uint4((uint32_t)i, 1, 2, 3);
    }
}

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

#include "ubench_variants.hpp"

int main(int argc, char ** argv)
{
    int n_frames = argc > 1 ? atoi(argv[1]) : 4096;
    const int W = argc > 2 ? atoi(argv[2]) : 480, H = argc > 3 ? atoi(argv[3]) : 320;
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
        printf("%-44s %8.4f ms  %8.1f GB/s  %5.1f%% of 8 TB/s  %9.1f Mpx/s\n", name, ms, bytes / ms / 1e6,
               bytes / ms / 1e6 / 80.0, px / ms / 1e3);
        fflush(stdout);
    };

    {   // ceiling: copy src(4B/px) -> ... use the 4 B/px buffer as both streams' source to move 8 B/px... keep simple:
        size_t n16 = px * 2 / 16;  // dst -> out, 2R + 2W per px
        float ms = T.run([&] { hipLaunchKernelGGL((k_copy<4, false>), dim3((n16 + 1023) / 1024), dim3(256), 0, 0, (const uint4 *)dst, (uint4 *)out, n16); });
        report("copy u16 frames (unroll 4)", px * 4.0, ms);
        ms = T.run([&] { hipLaunchKernelGGL((k_copy<8, false>), dim3((n16 + 2047) / 2048), dim3(256), 0, 0, (const uint4 *)dst, (uint4 *)out, n16); });
        report("copy u16 frames (unroll 8)", px * 4.0, ms);
        ms = T.run([&] { hipLaunchKernelGGL((k_copy<4, true>), dim3((n16 + 1023) / 1024), dim3(256), 0, 0, (const uint4 *)dst, (uint4 *)out, n16); });
        report("copy u16 frames (unroll 4, nontemporal)", px * 4.0, ms);
        size_t n32 = px * 4 / 16;
        ms = T.run([&] { hipLaunchKernelGGL((k_read<8>), dim3((n32 + 2047) / 2048), dim3(256), 0, 0, (const uint4 *)src, (uint4 *)out, n32); });
        report("read-only 4 B/px stream", px * 4.0, ms);
        ms = T.run([&] { hipLaunchKernelGGL((k_write<4>), dim3((n16 + 1023) / 1024), dim3(256), 0, 0, (uint4 *)out, n16); });
        report("write-only 2 B/px stream", px * 2.0, ms);
    }

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
    {   // production kernels
        const uint32_t gpf = (uint32_t)((size_t)W * H / 8), cpf = (gpf + LIN_CHUNK - 1) / LIN_CHUNK;
        float ms = T.run([&] { hipLaunchKernelGGL(k_blend_linear<SRC_ARGB8888>, dim3(cpf * n_frames), dim3(256), 0, 0, a, gpf, cpf); });
        report("k_blend_linear<ARGB8888> (8 B/px)", px * 8.0, ms);
        a.out = out;
        a.out_fs = (long long)W * H;
        a.rot = 0;
        ms = T.run([&] { hipLaunchKernelGGL(k_blend_linear<SRC_ARGB8888>, dim3(cpf * n_frames), dim3(256), 0, 0, a, gpf, cpf); });
        report("k_blend_linear<ARGB8888>+copy out (10 B/px)", px * 10.0, ms);
        a.rot = 1;
        const uint32_t tiles_x = (W + 63) / 64, tiles_y = (H + 63) / 64, tpf = tiles_x * tiles_y;
        const uint32_t n_tiles = tpf * n_frames;
        ms = T.run([&] { hipLaunchKernelGGL((k_tile<SRC_ARGB8888, 8, false>), dim3((n_tiles + TILE_WPB - 1) / TILE_WPB), dim3(64 * TILE_WPB), 0, 0, a, W, H, tiles_x, tiles_y, tpf, n_tiles); });
        report("k_tile<ARGB8888> fused rot90 (10 B/px)", px * 10.0, ms);
        ms = T.run([&] { hipLaunchKernelGGL((k_tile<SRC_COPY, 8, false>), dim3((n_tiles + TILE_WPB - 1) / TILE_WPB), dim3(64 * TILE_WPB), 0, 0, a, W, H, tiles_x, tiles_y, tpf, n_tiles); });
        report("k_tile<COPY> rot90 (4 B/px)", px * 4.0, ms);
    }
    {   // the column kernel on this frame size, 5 / 8 / 10 wavefronts per workgroup (5: a workgroup writes 64 whole,
        // consecutive panel rows = 40 KB contiguous).  480x320: fused 61 / 64 / 56 %, rotation alone 71 / 68 / 59 %;
        // k_tile: 70 / 74 % -- contiguous output is not what the fused kernel is short of
        a.rot = 1;
        char name[96];
        for(int wpb : {5, 8, 10}) {
            const uint32_t tiles_x = (W + TILE - 1) / TILE, cols_y = (H + TILE * wpb - 1) / (TILE * wpb);
            const uint32_t n_blocks = tiles_x * cols_y * n_frames;
            const uint32_t lds = 8192u * wpb;
            snprintf(name, sizeof name, "k_tile_col<ARGB8888> WPB %d fused (10 B/px)", wpb);
            float ms = T.run([&] { hipLaunchKernelGGL((k_tile_col<SRC_ARGB8888, 4, false>), dim3(n_blocks), dim3(64 * wpb), lds, 0, a, W, H, tiles_x, cols_y, wpb, 1); });
            report(name, px * 10.0, ms);
            snprintf(name, sizeof name, "k_tile_col<COPY> WPB %d rot90 (4 B/px)", wpb);
            ms = T.run([&] { hipLaunchKernelGGL((k_tile_col<SRC_COPY, 4, false>), dim3(n_blocks), dim3(64 * wpb), lds, 0, a, W, H, tiles_x, cols_y, wpb, 1); });
            report(name, px * 4.0, ms);
        }
    }
    if(getenv("UBENCH_VARIANTS")) run_variants(a, W, H, n_frames, px, T, report);
    return 0;
}
