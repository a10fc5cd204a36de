// Issue rate of the integer multiply forms a pixel mix can be built from, on every SIMD at once.
// Prints wave-instructions per second per form; full rate on MI355X is about 614 G/s (1024 SIMDs x 2.4 GHz / 4).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t * out, uint32_t seed, int iters)
{
    uint32_t a[8];
#pragma unroll
    for(int i = 0; i < 8; i++) a[i] = seed + threadIdx.x * 8 + i;
    uint32_t b = seed | 3u;
    for(int it = 0; it < iters; it++) {
#pragma unroll
        for(int r = 0; r < 8; r++) {
#pragma unroll
            for(int i = 0; i < 8; i++) {
                if(OP == 0) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if(OP == 1) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if(OP == 2) asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if(OP == 3) asm volatile("v_pk_mad_u16 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b));
                if(OP == 4) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b));
                if(OP == 5) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if(OP == 6) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if(OP == 7) asm volatile("v_pk_lshrrev_b16 %0, 3, %0" : "+v"(a[i]));
                if(OP == 8) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
                if(OP == 9) asm volatile("v_and_or_b32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b));
            }
        }
    }
    uint32_t s = 0;
#pragma unroll
    for(int i = 0; i < 8; i++) s ^= a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP>
void run(const char * name, uint32_t * out)
{
    const int blocks = 256 * 8, iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<OP><<<blocks, 256>>>(out, 1, 10);
    hipEventRecord(e0);
    k<OP><<<blocks, 256>>>(out, 1, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double winstr = (double)blocks * 4 * iters * 64;
    printf("%-18s %8.1f G wave-instr/s  (%.3f ms)\n", name, winstr / ms / 1e6, ms);
}

int main()
{
    uint32_t * out;
    if(hipMalloc(&out, 256 * 8 * 256 * 4) != hipSuccess) return 1;
    run<5>("v_add_u32", out);
    run<0>("v_mul_lo_u32", out);
    run<1>("v_mul_u32_u24", out);
    run<4>("v_mad_u32_u24", out);
    run<2>("v_pk_mul_lo_u16", out);
    run<3>("v_pk_mad_u16", out);
    run<6>("v_pk_add_u16", out);
    run<7>("v_pk_lshrrev_b16", out);
    run<8>("v_perm_b32", out);
    run<9>("v_and_or_b32", out);
    hipFree(out);
    return 0;
}
