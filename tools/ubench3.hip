// ubench3.hip -- development microbenchmark: cache-policy bits on streaming loads and stores (plain, nt, sc1, sc0 sc1)
// for a device copy and for the 6 R : 4 W mix of the fused path.  Not part of the product or the tests.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/ubench3 tools/ubench3.hip && ./tools/ubench3
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <vector>

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e = (x);                                                            \
        if(e != hipSuccess) {                                                          \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e)); \
            exit(1);                                                                   \
        }                                                                              \
    } while(0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int P>
__device__ __forceinline__ u32x4 ld(const u32x4 * p)
{
    u32x4 v;
    if(P == 0) v = *p;
    else if(P == 1) asm volatile("global_load_dwordx4 %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if(P == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <int P>
__device__ __forceinline__ void st(u32x4 * p, u32x4 v)
{
    if(P == 0) *p = v;
    else if(P == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
    else if(P == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    else if(P == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(p), "v"(v) : "memory");
}

// out[i] = in[i]; UN 16-byte words per lane in flight (policy loads wait one by one: compare store policies with LP = 0)
template <int LP, int SP, int UN>
__global__ __launch_bounds__(256) void k_copy(const u32x4 * __restrict__ in, u32x4 * __restrict__ out, size_t n)
{
    const size_t base = (size_t)blockIdx.x * 256 * UN + threadIdx.x;
    u32x4 v[UN];
#pragma unroll
    for(int u = 0; u < UN; u++)
        if(base + (size_t)u * 256 < n) v[u] = ld<LP>(in + base + (size_t)u * 256);
#pragma unroll
    for(int u = 0; u < UN; u++)
        if(base + (size_t)u * 256 < n) st<SP>(out + base + (size_t)u * 256, v[u]);
}

// the fused mix, linear: a (4 B/px) read, b (2 B/px) read-modify-written, c (2 B/px) written
template <int SPB, int SPC>
__global__ __launch_bounds__(256) void k_mix(const u32x4 * __restrict__ a, u32x4 * __restrict__ b, u32x4 * __restrict__ c, size_t n)
{
    const size_t base = (size_t)blockIdx.x * 512 + threadIdx.x;
    u32x4 va[2][2], vb[2];
#pragma unroll
    for(int u = 0; u < 2; u++) {
        const size_t i = base + (size_t)u * 256;
        if(i < n) {
            vb[u] = b[i];
            va[u][0] = a[2 * i];
            va[u][1] = a[2 * i + 1];
        }
    }
#pragma unroll
    for(int u = 0; u < 2; u++) {
        const size_t i = base + (size_t)u * 256;
        if(i < n) {
            u32x4 r = vb[u] + va[u][0] * 3u + va[u][1];
            st<SPB>(b + i, r);
            st<SPC>(c + i, r);
        }
    }
}

struct Timer {
    hipEvent_t a, b;
    Timer()
    {
        CK(hipEventCreate(&a));
        CK(hipEventCreate(&b));
    }
    float run(const std::function<void()> & f, int warm = 3, int iters = 15)
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

int main()
{
    const size_t px = (size_t)480 * 320 * 4096;
    const size_t n = px * 2 / 16;  // 16-byte words of a 2 B/px buffer
    u32x4 *a, *b, *c;
    CK(hipMalloc(&a, px * 4));
    CK(hipMalloc(&b, px * 2));
    CK(hipMalloc(&c, px * 2));
    CK(hipMemset(a, 1, px * 4));
    CK(hipMemset(b, 2, px * 2));
    Timer T;
    auto report = [&](const char * name, double bytes, float ms) {
        printf("%-64s %8.4f ms  %8.1f GB/s  %5.1f%%\n", name, ms, bytes / ms / 1e6, bytes / ms / 1e6 / 80.0);
        fflush(stdout);
    };
    const char * sp[] = {"plain", "nt", "sc1", "sc0 sc1", "sc1 nt"};
    char nm[128];
#define COPY(LP, SP, UN, PAD)                                                                                                   \
    {                                                                                                                           \
        float ms = T.run([&] { hipLaunchKernelGGL((k_copy<LP, SP, UN>), dim3((n + 256 * UN - 1) / (256 * UN)), dim3(256), PAD, 0, b, c, n); }); \
        snprintf(nm, sizeof nm, "copy: loads %s, stores %s, %d words/lane, LDS pad %d", sp[LP], sp[SP], UN, PAD);               \
        report(nm, px * 4.0, ms);                                                                                               \
    }
    COPY(0, 0, 2, 0) COPY(0, 0, 2, 32768) COPY(0, 0, 4, 0) COPY(0, 0, 4, 32768) COPY(0, 0, 8, 0)
    COPY(0, 1, 2, 0) COPY(0, 1, 2, 32768) COPY(0, 2, 2, 0) COPY(0, 2, 2, 32768) COPY(0, 3, 2, 0) COPY(0, 3, 2, 32768) COPY(0, 4, 2, 32768)
    COPY(1, 0, 1, 0) COPY(1, 0, 1, 32768) COPY(2, 0, 1, 0) COPY(2, 0, 1, 32768) COPY(0, 0, 1, 0) COPY(0, 0, 1, 32768)
#define MIX(SPB, SPC, PAD)                                                                                                      \
    {                                                                                                                           \
        float ms = T.run([&] { hipLaunchKernelGGL((k_mix<SPB, SPC>), dim3((n + 511) / 512), dim3(256), PAD, 0, a, b, c, n); }); \
        snprintf(nm, sizeof nm, "mix 6R:4W: RMW stores %s, out stores %s, LDS pad %d", sp[SPB], sp[SPC], PAD);                  \
        report(nm, px * 10.0, ms);                                                                                              \
    }
    MIX(0, 0, 0) MIX(0, 0, 49152) MIX(0, 1, 49152) MIX(0, 2, 49152) MIX(0, 3, 49152) MIX(0, 4, 49152) MIX(1, 1, 49152) MIX(2, 2, 49152) MIX(2, 1, 49152)
    return 0;
}
