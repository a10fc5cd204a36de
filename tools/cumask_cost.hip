// Cost of asking a stream for its CU mask (host call), per call.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
int main()
{
    hipStream_t st;
    if(hipStreamCreate(&st) != hipSuccess) return 1;
    uint32_t mask[16];
    int n = 0;
    auto t0 = std::chrono::steady_clock::now();
    for(int i = 0; i < 100000; i++) {
        if(hipExtStreamGetCUMask(i & 1 ? st : nullptr, 16, mask) != hipSuccess) return 2;
        n += __builtin_popcount(mask[0]);
    }
    double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 100000;
    int bits = 0;
    for(int i = 0; i < 16; i++) bits += __builtin_popcount(mask[i]);
    printf("hipExtStreamGetCUMask: %.3f us per call, %d CUs in the mask of an ordinary stream (%d)\n", us, bits, n & 1);
    return 0;
}
