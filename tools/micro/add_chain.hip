// Microbenchmark: cycles per dependent v_add_f64 on one lane / one wave of an otherwise idle CU, operands in registers
// or streamed from LDS (the shape of the ordered sums in lsq_spmd.hpp).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void chain_regs(double *out, unsigned long long *cyc, int reps, int active)
{
    double v[16];
    for (int q = 0; q < 16; ++q) v[q] = 1.0 + q * 1e-3 + threadIdx.x;
    double t = 0.0;
    if ((int)threadIdx.x < active) {
        const unsigned long long t0 = __builtin_readcyclecounter();
        for (int r = 0; r < reps; ++r) {
#pragma unroll
            for (int q = 0; q < 16; ++q) t = t + v[q];
        }
        const unsigned long long t1 = __builtin_readcyclecounter();
        if (threadIdx.x == 0) cyc[0] = t1 - t0;
    }
    out[threadIdx.x] = t;
}
__global__ void chain_lds(double *out, unsigned long long *cyc, int count, int reps)
{
    __shared__ double p[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) p[i] = 1.0 + i * 1e-6;
    __syncthreads();
    double t = 0.0;
    if (threadIdx.x == 0) {
        const unsigned long long t0 = __builtin_readcyclecounter();
        for (int r = 0; r < reps; ++r) {
            for (int j = 0; j + 16 <= count; j += 16) {
                double v[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) v[q] = p[j + q + 1];
#pragma unroll
                for (int q = 0; q < 16; ++q) t = t + v[q];
            }
        }
        const unsigned long long t1 = __builtin_readcyclecounter();
        cyc[0] = t1 - t0;
    }
    out[threadIdx.x] = t;
}
int main()
{
    double *out; unsigned long long *cyc, h;
    hipMalloc(&out, 1024 * 8); hipMalloc(&cyc, 8);
    for (int active : {1, 64}) {
        hipLaunchKernelGGL(chain_regs, dim3(1), dim3(64), 0, 0, out, cyc, 1000, active);
        hipLaunchKernelGGL(chain_regs, dim3(1), dim3(64), 0, 0, out, cyc, 1000, active);
        hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        printf("registers, %2d lanes active: %.2f cycles per dependent add\n", active, (double)h / 16000.0);
    }
    for (int count : {32, 112, 1024}) {
        hipLaunchKernelGGL(chain_lds, dim3(1), dim3(256), 0, 0, out, cyc, count, 200);
        hipLaunchKernelGGL(chain_lds, dim3(1), dim3(256), 0, 0, out, cyc, count, 200);
        hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        printf("LDS blocks of 16, %4d terms per sum: %.2f cycles per term\n", count, (double)h / (200.0 * (count / 16 * 16)));
    }
    return 0;
}
