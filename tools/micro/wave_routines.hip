// Microbenchmark: cycles per call of the single-wavefront recurrences of lsq_wave.hpp on an otherwise idle CU
// (n = 120, operands in LDS as in the QP kernel).
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../mpc_amd/csrc/lsq_wave.hpp"
#ifndef NN
#define NN 120
#endif
constexpr int N = NN, TRI = N * (N + 1) / 2 + 1;
__global__ void __launch_bounds__(256) bench(unsigned long long *cyc, double *out, int reps)
{
    __shared__ double E[TRI], x[N], g[N], ff[N], vv[N], ys[2 * N];
    for (int i = threadIdx.x; i < TRI; i += blockDim.x) E[i] = 0.001 * ((i * 37) % 101) - 0.05;
    __syncthreads();
    for (int i = threadIdx.x; i < N; i += blockDim.x) {
        E[i * N - (i * (i - 1)) / 2] = 1.5 + 0.01 * i;
        x[i] = 1.0 + 0.1 * i; g[i] = 0.5 - 0.01 * i;
    }
    __syncthreads();
#if defined(__HIP_DEVICE_COMPILE__)
    if (threadIdx.x < 64) {
        unsigned long long t0 = __builtin_readcyclecounter();
        for (int r = 0; r < reps; ++r) lsqwave::back_substitute(E, x, ys, N);
        unsigned long long t1 = __builtin_readcyclecounter();
        if (threadIdx.x == 0) cyc[0] = t1 - t0;
        t0 = __builtin_readcyclecounter();
        for (int r = 0; r < reps; ++r) lsqwave::forward_substitute(E, g, ff, ys, N);
        t1 = __builtin_readcyclecounter();
        if (threadIdx.x == 0) cyc[1] = t1 - t0;
        t0 = __builtin_readcyclecounter();
        for (int r = 0; r < reps; ++r) lsqwave::unit_lower_solve(E, g, vv, N);
        t1 = __builtin_readcyclecounter();
        if (threadIdx.x == 0) cyc[2] = t1 - t0;
    }
#endif
    __syncthreads();
    if (threadIdx.x < N) out[threadIdx.x] = x[threadIdx.x] + ff[threadIdx.x] + vv[threadIdx.x];
}
int main()
{
    unsigned long long *cyc, h[3]; double *out;
    (void)hipMalloc(&cyc, 24); (void)hipMalloc(&out, N * 8);
    const int reps = 20;
    for (int pass = 0; pass < 2; ++pass) {
        hipLaunchKernelGGL(bench, dim3(1), dim3(256), 0, 0, cyc, out, reps);
        (void)hipMemcpy(h, cyc, 24, hipMemcpyDeviceToHost);
    }
    printf("back_substitute    %8.0f cycles per call (%.1f per term of the chain)\n", (double)h[0] / reps, (double)h[0] / reps / (N * (N - 1) / 2));
    printf("forward_substitute %8.0f cycles per call (%.0f per step)\n", (double)h[1] / reps, (double)h[1] / reps / (N - 1));
    printf("unit_lower_solve   %8.0f cycles per call (%.0f per step)\n", (double)h[2] / reps, (double)h[2] / reps / (N - 1));
    return 0;
}
