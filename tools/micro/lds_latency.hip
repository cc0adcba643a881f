// Microbenchmark: LDS round trip of one wavefront on an idle CU (dependent ds_read_b32 chain), and the cost of a
// ds_read_b64 whose result is needed k independent FP64 operations later.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void chase(int *out, unsigned long long *cyc, int reps)
{
    __shared__ int nxt[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) nxt[i] = (i * 17 + 5) & 1023;
    __syncthreads();
    int p = threadIdx.x;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int q = 0; q < 16; ++q) p = nxt[p];
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
    out[threadIdx.x] = p;
}
template <int GAP>
__global__ void gap(double *out, unsigned long long *cyc, int reps)
{
    __shared__ double a[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) a[i] = 1e-3 * (i % 13);
    __syncthreads();
    double t = 1.0 + threadIdx.x * 1e-3, c = a[threadIdx.x];
    int at = threadIdx.x;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            at = (at + 64) & 4095;
            const double nc = a[at];                    // needed after GAP dependent operations
#pragma unroll
            for (int g = 0; g < GAP; ++g) t = t + t * c;
            c = nc;
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
    out[threadIdx.x] = t;
}
int main()
{
    int *oi; double *od; unsigned long long *cyc, h;
    (void)hipMalloc(&oi, 256); (void)hipMalloc(&od, 512); (void)hipMalloc(&cyc, 8);
    const int reps = 500;
    for (int pass = 0; pass < 2; ++pass) {
        hipLaunchKernelGGL(chase, dim3(1), dim3(64), 0, 0, oi, cyc, reps);
        (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        if (pass) printf("dependent ds_read_b32: %.1f cycles each\n", (double)h / (16.0 * reps));
    }
#define RUN(G)                                                                              \
    for (int pass = 0; pass < 2; ++pass) {                                                  \
        hipLaunchKernelGGL(gap<G>, dim3(1), dim3(64), 0, 0, od, cyc, reps);                 \
        (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);                                 \
        if (pass) printf("load used after %2d mul+add pairs: %.1f cycles per load (%.1f without the load)\n", G, (double)h / (8.0 * reps), 8.5 * G); \
    }
    RUN(1) RUN(2) RUN(4) RUN(8) RUN(16)
    return 0;
}
