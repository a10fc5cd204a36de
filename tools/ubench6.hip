// ubench6.hip -- does it pay to shape HBM traffic in time?  (development tool, not part of the product or the tests)
// The fused blend + flush kernel moves 6 bytes in and 4 bytes out per pixel, reads and writes mixed at every instant.
// Read-only streams run faster than mixed ones on this device; this program measures what separating the directions
// could give at best, and at which phase length the gain is gone:
//   mixed        one kernel, every lane loads 3 x 16 bytes and stores 2 x 16 bytes (the 6 : 4 mix)
//   separated    a read-only kernel over all 6 parts, then a write-only kernel over all 4 (infinitely long phases)
//   alternating  the same bytes in chunks: read chunk, write chunk, read chunk, ... as kernels on one stream, the
//                kernel boundary standing in for the barrier a persistent kernel would need between its phases
//   concurrent   the read-only and the write-only kernel at the same time on two streams (no alternation at all)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/ubench6 tools/ubench6.hip && ./tools/ubench6 [MB per part]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                                \
    do {                                                                                     \
        hipError_t e = (x);                                                                  \
        if(e != hipSuccess) {                                                                \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e)); \
            exit(1);                                                                         \
        }                                                                                    \
    } while(0)

// blocks of one XCD take one contiguous eighth of the range (as the product kernels do)
__device__ __forceinline__ uint32_t xcd_remap(uint32_t b, uint32_t n)
{
    const uint32_t q = n >> 3, r = n & 7u, x = b & 7u;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}

constexpr int U = 4;  // 16-byte words per lane and stream
__global__ __launch_bounds__(256) void k_read(const uint4 * __restrict__ in, uint4 * __restrict__ sink, size_t n)
{
    const size_t base = (size_t)xcd_remap(blockIdx.x, gridDim.x) * 256 * U + threadIdx.x;
    uint4 acc = make_uint4(0, 0, 0, 0);
#pragma unroll
    for(int u = 0; u < U; u++) {
        const size_t i = base + (size_t)u * 256;
        if(i < n) {
            const uint4 v = in[i];
            acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
        }
    }
    if(acc.x == 0x12345678u && acc.y == 0x9abcdef0u) sink[0] = acc;  // never true for the test data; keeps the loads
}
__global__ __launch_bounds__(256) void k_write(uint4 * __restrict__ out, size_t n, uint32_t tag)
{
    const size_t base = (size_t)xcd_remap(blockIdx.x, gridDim.x) * 256 * U + threadIdx.x;
#pragma unroll
    for(int u = 0; u < U; u++) {
        const size_t i = base + (size_t)u * 256;
        if(i < n) out[i] = make_uint4(tag, (uint32_t)i, tag, tag);
    }
}
// n = words per part; part p of the input starts at in + p * n, part q of the output at out + q * n
__global__ __launch_bounds__(256) void k_mixed(const uint4 * __restrict__ in, uint4 * __restrict__ out, size_t n)
{
    const size_t base = (size_t)xcd_remap(blockIdx.x, gridDim.x) * 256 * 2 + threadIdx.x;
#pragma unroll
    for(int u = 0; u < 2; u++) {
        const size_t i = base + (size_t)u * 256;
        if(i < n) {
            const uint4 a = in[i], b = in[n + i], c = in[2 * n + i];
            out[i] = make_uint4(a.x ^ b.x, a.y ^ c.y, b.z, c.w);
            out[n + i] = make_uint4(a.w, b.y, c.x, a.z);
        }
    }
}

static dim3 grid_for(size_t n, int per_block) { return dim3((unsigned)((n + per_block - 1) / per_block)); }

int main(int argc, char ** argv)
{
    const size_t part_mb = argc > 1 ? (size_t)atoll(argv[1]) : 1024;   // the "pixel count": 2 bytes per pixel and part
    const size_t part = part_mb << 20, n = part / 16;
    uint4 *in, *out, *sink;
    CK(hipMalloc(&in, 3 * part));
    CK(hipMalloc(&out, 2 * part));
    CK(hipMalloc(&sink, 256));
    CK(hipMemset(in, 1, 3 * part));
    CK(hipMemset(out, 0, 2 * part));
    hipStream_t s1, s2;
    CK(hipStreamCreate(&s1));
    CK(hipStreamCreate(&s2));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const double bytes = 5.0 * (double)part;
    auto report = [&](const char * name, float ms) {
        printf("%-58s %8.3f ms  %7.1f GB/s  %5.1f %% of 8 TB/s\n", name, ms, bytes / ms / 1e6, bytes / ms / 1e6 / 80.0);
        fflush(stdout);
    };
    auto timed = [&](auto fn) {
        fn();
        CK(hipDeviceSynchronize());
        float best = 1e30f;
        for(int r = 0; r < 3; r++) {
            CK(hipEventRecord(e0, s1));
            for(int k = 0; k < 5; k++) fn();
            CK(hipEventRecord(e1, s1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms / 5 < best ? ms / 5 : best;
        }
        return best;
    };
    report("mixed (3 words in, 2 out per lane step)", timed([&] { hipLaunchKernelGGL(k_mixed, grid_for(n, 512), dim3(256), 0, s1, in, out, n); }));
    report("separated (read all, then write all)", timed([&] {
               hipLaunchKernelGGL(k_read, grid_for(3 * n, 256 * U), dim3(256), 0, s1, in, sink, 3 * n);
               hipLaunchKernelGGL(k_write, grid_for(2 * n, 256 * U), dim3(256), 0, s1, out, 2 * n, 7u);
           }));
    for(size_t chunk_mb : {256, 64, 16, 8, 4, 2}) {   // megabytes of the 5 parts handled per read + write pair
        const size_t cw = (chunk_mb << 20) / 16 / 5;     // words per part and chunk
        if(cw == 0 || cw > n) continue;
        char name[96];
        snprintf(name, sizeof name, "alternating, %3zu MB per phase pair (%5.1f us at 5.6 TB/s)", chunk_mb, (double)(chunk_mb << 20) / 5.6e6);
        report(name, timed([&] {
                   for(size_t at = 0; at < n; at += cw) {
                       const size_t m = n - at < cw ? n - at : cw;
                       for(int p = 0; p < 3; p++)   // the three input parts of this chunk, one launch each keeps the grid simple
                           hipLaunchKernelGGL(k_read, grid_for(m, 256 * U), dim3(256), 0, s1, in + p * n + at, sink, m);
                       for(int q = 0; q < 2; q++)
                           hipLaunchKernelGGL(k_write, grid_for(m, 256 * U), dim3(256), 0, s1, out + q * n + at, m, 9u);
                   }
               }));
    }
    {   // readers and writers at the same time, unsynchronised
        hipEvent_t f2;
        CK(hipEventCreate(&f2));
        report("concurrent (read-only and write-only kernels, two streams)", timed([&] {
                   hipLaunchKernelGGL(k_write, grid_for(2 * n, 256 * U), dim3(256), 0, s2, out, 2 * n, 5u);
                   hipLaunchKernelGGL(k_read, grid_for(3 * n, 256 * U), dim3(256), 0, s1, in, sink, 3 * n);
                   CK(hipEventRecord(f2, s2));
                   CK(hipStreamWaitEvent(s1, f2, 0));
               }));
    }
    CK(hipGetLastError());
    CK(hipFree(in));
    CK(hipFree(out));
    CK(hipFree(sink));
    return 0;
}
