// ubench4.hip -- issue cost of the integer instructions the pixel arithmetic is made of (development tool).
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench4 tools/ubench4.hip && ./tools/ubench4
// Every wave runs ITER x 16 independent instructions of one kind; with all SIMDs loaded the time per instruction
// per SIMD is the instruction's throughput cost.  Measured at 4 waves per SIMD, cycles of the nominal 2.4 GHz clock:
//   v_add/sub/and/xor_u32, v_lshrrev_b32, v_mul_lo_u16                       2.5 - 2.9
//   v_bfe_u32, v_cvt_f32_u32, v_fma_f32, v_max_u32, v_cmp, v_cndmask          3.9 - 4.5
//   v_mul_u32_u24, v_mul_hi_u32_u24                                          4.4 - 4.7
//   v_mad_u32_u24, v_perm_b32, v_and_or_b32, v_lshl_or_b32, v_bfi_b32,
//   v_alignbit_b32, v_pk_mul_lo_u16, v_pk_mad_u16, SDWA forms, v_mul_hi_u32  4.9 - 5.0
//   v_mad_u64_u32 5.2, v_mul_lo_u32 5.5, v_rcp_f32 8.2, v_mad_u16 9.0
// A three-operand instruction costs about what the two simple ones it replaces cost; the full 32-bit multiply is no
// quarter-rate instruction here.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if(e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while(0)
constexpr int ITER = 4096;

#define KERNEL(name, INSTR)                                                                                   \
    __global__ __launch_bounds__(256) void name(uint32_t * out, uint32_t seed)                                \
    {                                                                                                         \
        uint32_t r[8];                                                                                        \
        _Pragma("unroll") for(int i = 0; i < 8; i++) r[i] = threadIdx.x * 7u + i + seed;                      \
        uint32_t b = seed | 3u;                                                                               \
        float fb = (float)b;                                                                                  \
        (void)fb;                                                                                             \
        for(int it = 0; it < ITER; it++) {                                                                    \
            _Pragma("unroll") for(int k = 0; k < 2; k++) {                                                    \
                _Pragma("unroll") for(int i = 0; i < 8; i++) { INSTR; }                                       \
            }                                                                                                 \
        }                                                                                                     \
        uint32_t s = 0;                                                                                       \
        _Pragma("unroll") for(int i = 0; i < 8; i++) s ^= r[i];                                               \
        if(s == 0x12345u) out[threadIdx.x] = s;                                                               \
    }
KERNEL(k_add, asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[i]) : "v"(b)))
KERNEL(k_mul_lo, asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r[i]) : "v"(b)))
KERNEL(k_mul_u24, asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(r[i]) : "v"(b)))
KERNEL(k_mad_u24, asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(r[i]) : "v"(b)))
KERNEL(k_mul_hi_u24, asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(r[i]) : "v"(b)))
KERNEL(k_mul_hi, asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(r[i]) : "v"(b)))
KERNEL(k_mad_u64, uint64_t t; asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(t) : "v"(r[i]), "v"(b) : "vcc"); r[i] = (uint32_t)t)
KERNEL(k_rcp, asm volatile("v_rcp_f32 %0, %0" : "+v"(r[i])))
KERNEL(k_cvt, asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(r[i])))
KERNEL(k_perm, asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(b)))
KERNEL(k_cndmask_indep, uint32_t t; asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(t) : "v"(r[i]), "v"(b)); r[(i + 3) & 7] ^= t)
KERNEL(k_xor, asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r[i]) : "v"(b)))
KERNEL(k_cndmask_e64, asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[10:11]" : "+v"(r[i]) : "v"(b) : "s10", "s11"))
KERNEL(k_cndmask_setvcc, asm volatile("v_cmp_lt_u32 vcc, %1, %0\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(b) : "vcc"))
KERNEL(k_cmp, asm volatile("v_cmp_lt_u32 vcc, %1, %0" :: "v"(r[i]), "v"(b) : "vcc"))
KERNEL(k_bfi, asm volatile("v_bfi_b32 %0, %1, %0, %1" : "+v"(r[i]) : "v"(b)))
KERNEL(k_max, asm volatile("v_max_u32 %0, %0, %1" : "+v"(r[i]) : "v"(b)))
KERNEL(k_lshl_or, asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(r[i]) : "v"(b)))
KERNEL(k_lshr, asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(r[i])))
KERNEL(k_and, asm volatile("v_and_b32 %0, %0, %1" : "+v"(r[i]) : "v"(b)))
KERNEL(k_sub, asm volatile("v_sub_u32 %0, %0, %1" : "+v"(r[i]) : "v"(b)))
KERNEL(k_fma, asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(b)))
KERNEL(k_alignbit, asm volatile("v_alignbit_b32 %0, %0, %1, 8" : "+v"(r[i]) : "v"(b)))
KERNEL(k_pk_mul_u16, asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(r[i]) : "v"(b)))
KERNEL(k_pk_mad_u16, asm volatile("v_pk_mad_u16 %0, %0, %1, %0" : "+v"(r[i]) : "v"(b)))
KERNEL(k_bfe, asm volatile("v_bfe_u32 %0, %0, 3, 8" : "+v"(r[i])))
KERNEL(k_and_or, asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(b)))
KERNEL(k_mul_u16, asm volatile("v_mul_lo_u16 %0, %0, %1" : "+v"(r[i]) : "v"(b)))
KERNEL(k_mad_u16, asm volatile("v_mad_u16 %0, %0, %1, %0" : "+v"(r[i]) : "v"(b)))
KERNEL(k_sdwa, asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "+v"(r[i]) : "v"(b)))

int main()
{
    uint32_t * out;
    CK(hipMalloc(&out, 4096));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    const double ghz = p.clockRate / 1e6;
    printf("%d CUs, %.2f GHz nominal\n", cus, ghz);
    auto run = [&](const char * name, void (*k)(uint32_t *, uint32_t)) {
        for(int wps = 2; wps <= 4; wps *= 2) {   // waves per SIMD
            dim3 grid(cus * wps), blk(256);
            hipLaunchKernelGGL(k, grid, blk, 0, 0, out, 1u);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(a));
            hipLaunchKernelGGL(k, grid, blk, 0, 0, out, 1u);
            CK(hipEventRecord(b));
            CK(hipEventSynchronize(b));
            float ms;
            CK(hipEventElapsedTime(&ms, a, b));
            const double instr_per_simd = (double)ITER * 16 * wps;
            printf("%-14s %d wave/SIMD: %7.3f ms  %5.2f cycles per wave-instruction per SIMD (at nominal clock)\n", name, wps, ms,
                   ms * 1e-3 * ghz * 1e9 / instr_per_simd);
        }
    };
    run("v_add_u32", k_add);
    run("v_mul_lo_u32", k_mul_lo);
    run("v_mul_u32_u24", k_mul_u24);
    run("v_mad_u32_u24", k_mad_u24);
    run("v_mul_hi_u24", k_mul_hi_u24);
    run("v_mul_hi_u32", k_mul_hi);
    run("v_mad_u64_u32", k_mad_u64);
    run("v_rcp_f32", k_rcp);
    run("v_cvt_f32_u32", k_cvt);
    run("v_perm_b32", k_perm);
    run("cndmask indep+xor", k_cndmask_indep);
    run("v_xor_b32", k_xor);
    run("cndmask e64 sgpr", k_cndmask_e64);
    run("cmp+cndmask (2)", k_cndmask_setvcc);
    run("v_cmp_lt_u32", k_cmp);
    run("v_bfi_b32", k_bfi);
    run("v_max_u32", k_max);
    run("v_lshl_or_b32", k_lshl_or);
    run("v_lshrrev_b32", k_lshr);
    run("v_and_b32", k_and);
    run("v_sub_u32", k_sub);
    run("v_fma_f32", k_fma);
    run("v_alignbit_b32", k_alignbit);
    run("v_pk_mul_lo_u16", k_pk_mul_u16);
    run("v_pk_mad_u16", k_pk_mad_u16);
    run("v_bfe_u32", k_bfe);
    run("v_and_or_b32", k_and_or);
    run("v_mul_lo_u16", k_mul_u16);
    run("v_mad_u16", k_mad_u16);
    run("v_add_sdwa", k_sdwa);
    return 0;
}
