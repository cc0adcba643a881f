// Check: lsq_wave.hpp's split division (divisor's reciprocal refined ahead, three operations once the dividend exists)
// gives the bits of num / den, on random operands whose exponents cover the band and a margin beyond it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../mpc_amd/csrc/lsq_wave.hpp"
__device__ unsigned long long mix(unsigned long long z)
{
    z += 0x9e3779b97f4a7c15ull; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
__device__ double make(unsigned long long r, int span)
{
    const unsigned long long mant = r & 0xfffffffffffffull, sign = (r >> 63) << 63;
    const long long e = 1023 + (long long)((r >> 52) % (unsigned)(2 * span + 1)) - span;
    return __longlong_as_double((long long)(sign | ((unsigned long long)e << 52) | mant));
}
__global__ void check(unsigned long long *bad, unsigned long long *fastcount, int rounds, int span)
{
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned long long s = mix(blockIdx.x * 1024ull + threadIdx.x + 77), nb = 0, nf = 0;
    for (int r = 0; r < rounds; ++r) {
        s = mix(s); const double n = make(s, span);
        s = mix(s); double d = make(s, span);
        if ((r & 15) == 3) d = __longlong_as_double(__double_as_longlong(d) | 0xfffffffffffffll);   // mantissa all ones
        if ((r & 15) == 7) d = __longlong_as_double(__double_as_longlong(d) & ~0xfffffffffffffll);  // a power of two
        const double y = lsqwave::recip_of(d);
        const double a = lsqwave::divide_by(n, d, y), b = n / d;
        nb += __double_as_longlong(a) != __double_as_longlong(b);
        nf += (y != 0.0 && lsqwave::in_band(n));
    }
    atomicAdd(bad, nb); atomicAdd(fastcount, nf);
#endif
}
int main()
{
    unsigned long long *d, h[2];
    (void)hipMalloc(&d, 16);
    for (int span : {3, 60, 390, 420, 1000}) {
        (void)hipMemset(d, 0, 16);
        hipLaunchKernelGGL(check, dim3(1024), dim3(256), 0, 0, d, d + 1, 1024, span);
        (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("exponents within +-%4d: %llu pairs, %llu through the split path, %llu differ\n", span, 1024ull * 256 * 1024, h[1], h[0]);
    }
    return 0;
}
