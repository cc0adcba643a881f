// Microbenchmark: latency of passing a double between lanes inside a dependent chain:
//   (a) two v_readlane_b32 into an SGPR pair that the next v_add_f64 reads,
//   (b) v_mov_b64 with DPP row_newbcast (inside a row of 16 lanes),
//   (c) ds_bpermute_b32 twice.
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ double bcast(double v, int lane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
__global__ void k_readlane(double *out, unsigned long long *cyc, int reps, int lane0)
{
    double t = 1.0 + threadIdx.x * 1e-3;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int q = 0; q < 16; ++q) t = t + bcast(t, (lane0 + q) & 63) * 1e-9;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
    out[threadIdx.x] = t;
}
__global__ void k_bperm(double *out, unsigned long long *cyc, int reps, int lane0)
{
    double t = 1.0 + threadIdx.x * 1e-3;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int a = ((lane0 + q) & 63) * 4;
            const int lo = __builtin_amdgcn_ds_bpermute(a, __double2loint(t));
            const int hi = __builtin_amdgcn_ds_bpermute(a, __double2hiint(t));
            t = t + __hiloint2double(hi, lo) * 1e-9;
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
    out[threadIdx.x] = t;
}
__global__ void k_dppmov(double *out, unsigned long long *cyc, int reps)
{
    double t = 1.0 + threadIdx.x * 1e-3;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            double b;
            asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "=v"(b) : "v"(t));
            t = t + b * 1e-9;
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
    out[threadIdx.x] = t;
}
__global__ void k_plain(double *out, unsigned long long *cyc, int reps)
{
    double t = 1.0 + threadIdx.x * 1e-3;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int q = 0; q < 16; ++q) t = t + t * 1e-9;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
    out[threadIdx.x] = t;
}
int main()
{
    double *out; unsigned long long *cyc, h;
    (void)hipMalloc(&out, 64 * 8); (void)hipMalloc(&cyc, 8);
    const int reps = 1000;
    for (int pass = 0; pass < 2; ++pass) {
        hipLaunchKernelGGL(k_plain, dim3(1), dim3(64), 0, 0, out, cyc, reps);
        (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        if (pass) printf("mul + add, same lane:            %.1f cycles per step\n", (double)h / (16.0 * reps));
        hipLaunchKernelGGL(k_readlane, dim3(1), dim3(64), 0, 0, out, cyc, reps, 3);
        (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        if (pass) printf("v_readlane x2 + mul + add:       %.1f cycles per step\n", (double)h / (16.0 * reps));
        hipLaunchKernelGGL(k_dppmov, dim3(1), dim3(64), 0, 0, out, cyc, reps);
        (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        if (pass) printf("v_mov_b64 row_newbcast + mul + add: %.1f cycles per step\n", (double)h / (16.0 * reps));
        hipLaunchKernelGGL(k_bperm, dim3(1), dim3(64), 0, 0, out, cyc, reps, 3);
        (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        if (pass) printf("ds_bpermute x2 + mul + add:      %.1f cycles per step\n", (double)h / (16.0 * reps));
    }
    return 0;
}
