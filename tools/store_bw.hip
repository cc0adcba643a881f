// Dev microbenchmark: how fast can waves stream per-wave-private 241,920-byte blocks to HBM?
// vy = 1: half of the doubles non-zero; vy = 0: all zeros (does the rate depend on the data?)
// All accesses are in bounds by construction: block b covers [b*BLK, (b+1)*BLK) doubles of a
// buffer of nblocks*BLK doubles.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double v2d __attribute__((ext_vector_type(2)));
constexpr int BLK = 120 * 252;  // doubles per block

// mode 0: flat aligned 1 KB stores; mode 1: 252-double columns as 64+62 lane stores;
// mode 2: flat, but consecutive blocks of the grid interleave chunks (grid-stride over the whole buffer)
template <int MODE>
__global__ __launch_bounds__(256) void k(double *out, int nblocks, int lds_pad, double vy)
{
    extern __shared__ double pad[];
    const int lane = threadIdx.x;
    if (lds_pad < 0) pad[lane] = 1.0;  // keep the LDS allocation alive
    const v2d z = {0.0, vy * (double)lane};
    if (MODE == 0) {
        double *base = out + (size_t)blockIdx.x * BLK;
        for (int c = 0; c < BLK / 128; ++c) *reinterpret_cast<v2d *>(base + 128 * c + 2 * lane) = z;
        const int rem = BLK - (BLK / 128) * 128;  // 32 doubles
        if (2 * lane < rem) *reinterpret_cast<v2d *>(base + (BLK / 128) * 128 + 2 * lane) = z;
    } else if (MODE == 1) {
        double *base = out + (size_t)blockIdx.x * BLK;
        for (int col = 0; col < 120; ++col) {
            double *dst = base + col * 252;
            *reinterpret_cast<v2d *>(dst + 2 * lane) = z;
            if (lane < 62) *reinterpret_cast<v2d *>(dst + 128 + 2 * lane) = z;
        }
    } else if (MODE == 3 || MODE == 4) {
        // 256-thread workgroup streams 4 consecutive blocks; all 4 waves work on the same block:
        // MODE 3: wave w takes columns [30w, 30w+30); MODE 4: wave w takes columns c = w (mod 4)
        const int wave = threadIdx.x >> 6, ln = threadIdx.x & 63;
        const v2d zz = {0.0, vy * (double)ln};
        for (int b = 0; b < 4; ++b) {
            double *base = out + ((size_t)blockIdx.x * 4 + b) * BLK;   // blockIdx.x < nblocks/4
            for (int i = 0; i < 30; ++i) {
                const int col = (MODE == 3) ? 30 * wave + i : 4 * i + wave;   // < 120
                double *dst = base + col * 252;
                *reinterpret_cast<v2d *>(dst + 2 * ln) = zz;
                if (ln < 62) *reinterpret_cast<v2d *>(dst + 128 + 2 * ln) = zz;
            }
        }
    } else if (MODE == 6 || MODE == 7) {
        // torch-fill-like: 256-thread block writes one contiguous 16 KB chunk (4 x 16 B per thread);
        // MODE 7: 64 KB chunk (16 x 16 B per thread)
        constexpr int U = (MODE == 6) ? 4 : 16;
        const size_t total = (size_t)nblocks * BLK / 2;                 // v2d elements
        v2d *o = reinterpret_cast<v2d *>(out);
        const size_t base = (size_t)blockIdx.x * 256 * U;
        const v2d zz = {0.0, vy};
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t i = base + (size_t)u * 256 + threadIdx.x;
            if (i < total) o[i] = zz;
        }
    } else if (MODE == 5) {
        // like MODE 1 with non-temporal stores
        double *base = out + (size_t)blockIdx.x * BLK;
        for (int col = 0; col < 120; ++col) {
            double *dst = base + col * 252;
            __builtin_nontemporal_store(z, reinterpret_cast<v2d *>(dst + 2 * lane));
            if (lane < 62) __builtin_nontemporal_store(z, reinterpret_cast<v2d *>(dst + 128 + 2 * lane));
        }
    } else {
        const size_t total = (size_t)nblocks * BLK / 2;  // v2d elements
        v2d *o = reinterpret_cast<v2d *>(out);
        for (size_t i = (size_t)blockIdx.x * 64 + lane; i < total; i += (size_t)gridDim.x * 64) o[i] = z;
    }
}

int main(int argc, char **argv)
{
    const int nblocks = 65536;
    double *buf;
    if (hipMalloc(&buf, sizeof(double) * (size_t)nblocks * BLK) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t bytes = sizeof(double) * (size_t)nblocks * BLK;
    for (double vy : {1.0, 0.0})
    for (int lds : {0, 20 * 1024}) {
        for (int mode = 0; mode < 8; ++mode) {
            float best = 1e9;
            for (int rep = 0; rep < 4; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(nblocks), dim3(64), lds, 0, buf, nblocks, 0, vy);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(nblocks), dim3(64), lds, 0, buf, nblocks, 0, vy);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256 * 8), dim3(64), lds, 0, buf, nblocks, 0, vy);
                if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(nblocks / 4), dim3(256), 3 * lds, 0, buf, nblocks, 0, vy);
                if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(nblocks / 4), dim3(256), 3 * lds, 0, buf, nblocks, 0, vy);
                if (mode == 6) hipLaunchKernelGGL(k<6>, dim3((unsigned)(((size_t)nblocks * BLK / 2 + 1023) / 1024)), dim3(256), 0, 0, buf, nblocks, 0, vy);
                if (mode == 7) hipLaunchKernelGGL(k<7>, dim3((unsigned)(((size_t)nblocks * BLK / 2 + 4095) / 4096)), dim3(256), 0, 0, buf, nblocks, 0, vy);
                if (mode == 5) hipLaunchKernelGGL(k<5>, dim3(nblocks), dim3(64), lds, 0, buf, nblocks, 0, vy);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            printf("vy=%.0f lds=%5d mode=%d: %.3f ms  %.0f GB/s\n", vy, lds, mode, best, bytes / best / 1e6);
        }
    }
    if (hipGetLastError() != hipSuccess) printf("HIP error\n");
    hipFree(buf);
    return 0;
}
