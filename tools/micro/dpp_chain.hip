// Microbenchmark: an ordered FP64 sum of 128 terms that one wavefront holds 16 lanes x 8 registers (each row of 16 lanes a copy),
// added with v_fmac_f64_dpp row_newbcast (t = p * 1.0 + t: the product is exact, so the one rounding is the addition's).
// Prints cycles per term and checks the bits against the serial sum.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>
#define TERM(r, c) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #c " row_mask:0xf bank_mask:0xf" : "+v"(t) : "v"(p[r]), "v"(one));
#define ROW(r) TERM(r, 0) TERM(r, 1) TERM(r, 2) TERM(r, 3) TERM(r, 4) TERM(r, 5) TERM(r, 6) TERM(r, 7) TERM(r, 8) TERM(r, 9) TERM(r, 10) TERM(r, 11) TERM(r, 12) TERM(r, 13) TERM(r, 14) TERM(r, 15)
__global__ void chain_dpp(double *out, unsigned long long *cyc, const double *in, int reps)
{
    double p[8];
    const int c = threadIdx.x & 15;
    for (int r = 0; r < 8; ++r) p[r] = in[c + 16 * r];
    double t = 0.0, one = 1.0;
    asm volatile("" : "+v"(one));
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int q = 0; q < reps; ++q) {
        ROW(0) ROW(1) ROW(2) ROW(3) ROW(4) ROW(5) ROW(6) ROW(7)
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
    out[threadIdx.x] = t;
}
int main()
{
    double *out, *in, h_in[128], h_out[64]; unsigned long long *cyc, h;
    hipMalloc(&out, 64 * 8); hipMalloc(&in, 128 * 8); hipMalloc(&cyc, 8);
    for (int i = 0; i < 128; ++i) h_in[i] = std::sin(1.0 + i) * std::pow(10.0, (i % 7) - 3);
    hipMemcpy(in, h_in, sizeof(h_in), hipMemcpyHostToDevice);
    for (int reps : {1, 1000}) {
        hipLaunchKernelGGL(chain_dpp, dim3(1), dim3(64), 0, 0, out, cyc, in, reps);
        hipLaunchKernelGGL(chain_dpp, dim3(1), dim3(64), 0, 0, out, cyc, in, reps);
        hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        hipMemcpy(h_out, out, sizeof(h_out), hipMemcpyDeviceToHost);
        volatile double t = 0.0;
        for (int q = 0; q < reps; ++q)
            for (int i = 0; i < 128; ++i) t = t + h_in[i];
        double ref = t;
        int same = 0;
        for (int l = 0; l < 64; ++l) same += std::memcmp(&h_out[l], &ref, 8) == 0;
        printf("reps %4d: %.2f cycles per term, %d of 64 lanes hold the serial sum's bits (%.17g vs %.17g)\n", reps, (double)h / (128.0 * reps), same, h_out[0], ref);
    }
    return 0;
}
