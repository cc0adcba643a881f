// Microbenchmark: what a predicate costs inside a dependent FP64 chain on one wavefront, by where the mask comes from.
#include <hip/hip_runtime.h>
#include <cstdio>
#define TIMED(name, ...)                                                                  \
    __global__ void name(double *out, unsigned long long *cyc, int reps, int k0)          \
    {                                                                                      \
        double t = 1.0 + threadIdx.x * 1e-3;                                               \
        const int lane = threadIdx.x;                                                      \
        const unsigned long long t0 = __builtin_readcyclecounter();                        \
        for (int r = 0; r < reps; ++r) {                                                   \
            _Pragma("unroll") for (int q = 0; q < 16; ++q) { const int k = (k0 + q + r) & 63; __VA_ARGS__ }   \
        }                                                                                  \
        const unsigned long long t1 = __builtin_readcyclecounter();                        \
        if (threadIdx.x == 0) cyc[0] = t1 - t0;                                            \
        out[threadIdx.x] = t;                                                              \
    }
TIMED(k_none, { (void)lane; (void)k; t = t + t * 1e-9; })
TIMED(k_vcmp, { const double u = t + t * 1e-9; t = lane > k ? u : t; })
TIMED(k_salu, {
    const double u = t + t * 1e-9;
    const unsigned long long m = ~0ull << ((k + 1) & 63);
    int lo = __double2loint(t), hi = __double2hiint(t);
    asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(lo) : "v"(__double2loint(u)), "s"(m));
    asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(hi) : "v"(__double2hiint(u)), "s"(m));
    t = __hiloint2double(hi, lo);
})
TIMED(k_branch, { const double u = t + t * 1e-9; if (__builtin_expect(((unsigned)__double2hiint(u) >> 20 & 0x7ff) - 623u <= 800u, 1)) t = u; else t = u / (double)(k + 3); })
int main()
{
    double *out; unsigned long long *cyc, h;
    (void)hipMalloc(&out, 64 * 8); (void)hipMalloc(&cyc, 8);
    const int reps = 1000;
#define RUN(kern, label)                                                                   \
    for (int pass = 0; pass < 2; ++pass) {                                                 \
        hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, 0, out, cyc, reps, 3);              \
        (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);                                \
        if (pass) printf("%-44s %.1f cycles per step\n", label, (double)h / (16.0 * reps)); \
    }
    RUN(k_none, "mul + add")
    RUN(k_vcmp, "mul + add + select (v_cmp in the loop)")
    RUN(k_salu, "mul + add + select (mask from the scalar unit)")
    RUN(k_branch, "mul + add + exponent check and branch")
    return 0;
}
