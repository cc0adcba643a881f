// Microbenchmark: which part of a column-oriented elimination step (lsq_wave.hpp: unit_lower_solve) costs what.
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ double bcast(double v, int lane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double select_lanes(double keep, double take, unsigned long long lanes)
{
    int lo = __double2loint(keep), hi = __double2hiint(keep);
    asm("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(lo) : "v"(__double2loint(take)), "s"(lanes));
    asm("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(hi) : "v"(__double2hiint(take)), "s"(lanes));
    return __hiloint2double(hi, lo);
}
template <bool LOADS, bool SELECT, bool BCAST>
__global__ void k(double *out, unsigned long long *cyc, int reps)
{
    __shared__ double a[64 * 64];
    for (int i = threadIdx.x; i < 64 * 64; i += 64) a[i] = 1e-3 * (i % 17);
    __syncthreads();
    const int lane = threadIdx.x;
    double v0 = 1.0 + lane * 1e-3;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
        const double *col = a;
        double c[4];
        c[0] = col[lane]; c[1] = col[64 + lane]; c[2] = col[128 + lane];
        col += 192;
        for (int i = 0; i < 60; i += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (LOADS) { c[(u + 3) & 3] = col[lane]; col += 64; }
                const double v = BCAST ? bcast(v0, i + u) : v0;
                const double nv = v0 - v * c[u];
                v0 = SELECT ? select_lanes(v0, nv, ~0ull << (i + u + 1)) : nv;
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
    out[threadIdx.x] = v0;
}
int main()
{
    double *out; unsigned long long *cyc, h;
    (void)hipMalloc(&out, 64 * 8); (void)hipMalloc(&cyc, 8);
    const int reps = 200;
#define RUN(L, S, B, label)                                                                \
    for (int pass = 0; pass < 2; ++pass) {                                                 \
        hipLaunchKernelGGL((k<L, S, B>), dim3(1), dim3(64), 0, 0, out, cyc, reps);         \
        (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);                                \
        if (pass) printf("%-40s %.1f cycles per step\n", label, (double)h / (60.0 * reps)); \
    }
    RUN(false, false, false, "mul + sub")
    RUN(false, false, true, "+ readlane")
    RUN(false, true, true, "+ readlane + select")
    RUN(true, false, false, "+ loads")
    RUN(true, true, true, "+ loads + readlane + select")
    return 0;
}
