// Microbenchmark: cycles of the QP's row transform (lsq_spmd.hpp: transform_rows, wavefront version) for ONE wavefront
// alone on a CU, per kind of row (a position row walks every block of e, a late bound row only the last ones).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../mpc_amd/csrc/lsq_spmd.hpp"
constexpr int H = 20, N = 6 * H, M = 12 * (H + 1), MG = M + 2 * N, TRI = N * (N + 1) / 2 + 1;
__global__ void __launch_bounds__(64) walk(unsigned long long *cyc, double *AT, const double *cvals, const double *nz, const double *lo,
                                           const double *hi, int p0, int reps)
{
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ double E[TRI], ff[N];
    for (int i = threadIdx.x; i < TRI; i += 64) E[i] = 0.001 * ((i * 37) % 101) - 0.05;
    __syncthreads();
    for (int i = threadIdx.x; i < N; i += 64) { E[i * N - (i * (i - 1)) / 2] = 1.5 + 0.01 * i; ff[i] = 0.3 - 0.001 * i; }
    __syncthreads();
    lsqspmd::Work W{};
    W.E = E; W.ff = ff; W.AT = AT;
    int rows[2];
    double *dst[2];
    for (int r = 0; r < 2; ++r) { rows[r] = lsqspmd::work_order(H, p0 + r * 64 + (int)threadIdx.x); dst[r] = AT + rows[r]; }
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int q = 0; q < reps; ++q) lsqspmd::transform_rows<2, true>(W, H, rows, cvals, nz, lo, hi, dst, true, AT);
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
#endif
}
int main()
{
    unsigned long long *cyc, h; double *AT, *cv, *nz, *lo, *hi;
    const int nzc = 6 * H + 3 * H * (H + 1);
    (void)hipMalloc(&cyc, 8); (void)hipMalloc(&AT, sizeof(double) * (N + 1) * MG); (void)hipMalloc(&cv, 8 * M);
    (void)hipMalloc(&nz, 8 * nzc); (void)hipMalloc(&lo, 8 * N); (void)hipMalloc(&hi, 8 * N);
    std::vector<double> v(nzc, 0.07);
    (void)hipMemcpy(nz, v.data(), 8 * nzc, hipMemcpyHostToDevice);
    (void)hipMemset(cv, 0, 8 * M); (void)hipMemset(lo, 0, 8 * N); (void)hipMemset(hi, 0, 8 * N); (void)hipMemset(AT, 0, sizeof(double) * (N + 1) * MG);
    const int reps = 5;
    for (int p0 : {0, 128, 256, 364}) {
        for (int pass = 0; pass < 2; ++pass) {
            hipLaunchKernelGGL(walk, dim3(1), dim3(64), 0, 0, cyc, AT, cv, nz, lo, hi, p0, reps);
            (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        }
        printf("rows %3d.. of the work order: %8.0f cycles per transform (%.1f per entry of a full walk)\n", p0, (double)h / reps, (double)h / reps / 7140.0);
    }
    return 0;
}
