// Dev microbenchmark: streaming-fill rates for the long-horizon expansion's store pattern.
// Buffer = nruns runs of RUN doubles; every index is checked against the run / buffer length.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v2d __attribute__((ext_vector_type(2)));

// mode 0: one workgroup per run, 16 B per lane per pass
// mode 1: one workgroup per run, 2 x 16 B per lane per pass (two passes unrolled)
// mode 2: persistent grid, workgroups stride over runs
// mode 3: flat grid-stride over the whole buffer (torch-fill-like)
template <int MODE>
__global__ void k(double *out, int nruns, int run)
{
    const int T = blockDim.x;
    const v2d z = {0.0, (double)threadIdx.x};
    if (MODE == 0) {
        double *base = out + (size_t)blockIdx.x * run;
        for (int e = 2 * threadIdx.x; e + 1 < run; e += 2 * T) *reinterpret_cast<v2d *>(base + e) = z;
    } else if (MODE == 1) {
        double *base = out + (size_t)blockIdx.x * run;
        for (int e = 2 * threadIdx.x; e + 1 < run; e += 4 * T) {
            *reinterpret_cast<v2d *>(base + e) = z;
            if (e + 2 * T + 1 < run) *reinterpret_cast<v2d *>(base + e + 2 * T) = z;
        }
    } else if (MODE == 2) {
        for (int r = blockIdx.x; r < nruns; r += gridDim.x) {
            double *base = out + (size_t)r * run;
            for (int e = 2 * threadIdx.x; e + 1 < run; e += 2 * T) *reinterpret_cast<v2d *>(base + e) = z;
        }
    } else {
        const size_t total = (size_t)nruns * run / 2;
        v2d *o = reinterpret_cast<v2d *>(out);
        for (size_t i = (size_t)blockIdx.x * T + threadIdx.x; i < total; i += (size_t)gridDim.x * T) o[i] = z;
    }
}

int main()
{
    const int run = 6 * 1212;                  // six columns of H=100
    const int nruns = 8192 * 100 / 4;          // 11.9 GB
    double *buf;
    const size_t bytes = sizeof(double) * (size_t)nruns * run;
    if (hipMalloc(&buf, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int threads[] = {64, 256, 384, 512, 1024};
    for (int mode = 0; mode < 4; ++mode)
        for (int t : threads) {
            float best = 1e9;
            for (int rep = 0; rep < 4; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(nruns), dim3(t), 0, 0, buf, nruns, run);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(nruns), dim3(t), 0, 0, buf, nruns, run);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256 * (2048 / t)), dim3(t), 0, 0, buf, nruns, run);
                if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(256 * (2048 / t)), dim3(t), 0, 0, buf, nruns, run);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            printf("mode=%d threads=%4d: %.3f ms  %.0f GB/s\n", mode, t, best, bytes / best / 1e6);
        }
    if (hipGetLastError() != hipSuccess) printf("HIP error\n");
    hipFree(buf);
    return 0;
}
