// Dev microbenchmark (round 3): what a zero-fill of the dense dc/dx buffer (65,536 x 120 x 252 doubles = 15.85 GB)
// can reach on this box, and what scattering the 1,380 non-zeros per scenario into it afterwards costs.
//   fill variants: hipMemsetAsync; short-lived workgroups of T threads writing U x 16 B per thread (block-contiguous);
//                  one 64-lane workgroup per 242 KB scenario run (the fused kernel's phase-3 pattern)
//   scatter:       one wavefront per scenario, 23 iterations x 60 lanes, one 8-byte store each at the dense position of
//                  quotient (j, a, k) — or the whole aligned 32-byte sector around it
// Every index is bounded by the buffer length.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef double v2d __attribute__((ext_vector_type(2)));
typedef double v4d __attribute__((ext_vector_type(4)));

template <int U>
__global__ void fill_block(double *out, size_t pairs)
{
    const size_t base = (size_t)blockIdx.x * blockDim.x * U + threadIdx.x;
    v2d *o = reinterpret_cast<v2d *>(out);
    const v2d z = {0.0, 0.0};
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const size_t i = base + (size_t)u * blockDim.x;
        if (i < pairs) o[i] = z;
    }
}

__global__ void fill_run(double *out, int run)   // one workgroup per run, lanes stride through it
{
    v2d *o = reinterpret_cast<v2d *>(out + (size_t)blockIdx.x * run);
    const v2d z = {0.0, 0.0};
    for (int e = threadIdx.x; e < run / 2; e += blockDim.x) o[e] = z;
}

// H = 20: lane = (pair pi of steps (pi, 19 - pi), joint a); walks 23 iterations like phase 2 of prefix_kernel
template <int WIDE>
__global__ void scatter(double *out, int count)
{
    const int H = 20, n = 120, m = 252;
    const int s = blockIdx.x;
    if (s >= count) return;
    const int it = threadIdx.x;
    if (it >= 60) return;
    const int pi = it / 6, a = it - 6 * pi;
    double *J = out + (size_t)s * n * m;
    int pert = 0, j = pi, k = pi;
    for (int step = 0; step < H + 3; ++step) {
        if (pert >= 2) break;
        const int var = 6 * j + a;
        if (k < H) {
            double *dst = J + (size_t)var * m + 12 * (k + 1) + a;
            const double v = 0.07 + step;
            if (WIDE) {
                const size_t e = (size_t)(dst - out);
                v4d w = {0.0, 0.0, 0.0, 0.0};
                w[e & 3] = v;
                *reinterpret_cast<v4d *>(out + (e & ~(size_t)3)) = w;
            } else {
                *dst = v;
            }
            if (k == j) {
                double *dv = J + (size_t)var * m + 12 * (j + 1) + 6 + a;
                if (WIDE) {
                    const size_t e = (size_t)(dv - out);
                    v4d w = {0.0, 0.0, 0.0, 0.0};
                    w[e & 3] = -1.0;
                    *reinterpret_cast<v4d *>(out + (e & ~(size_t)3)) = w;
                } else {
                    *dv = -1.0;
                }
            }
            ++k;
        } else {
            ++pert;
            j = H - 1 - pi;
            k = j;
        }
    }
}

// the library's fill-shaped expansion (kernels_prefix.hip: expand_flat_kernel, structured layout, H = 20) in stages:
// STAGE 0 stores only; 1 + index arithmetic; 2 + four loads of src[0] per lane; 3 + the real gather loads
struct FlatDiv { uint32_t mul, shift; };
static FlatDiv flat_div(uint32_t d)
{
    uint32_t L = 0;
    while ((1ull << L) < d) ++L;
    FlatDiv f;
    f.mul = (uint32_t)(((1ull << (31 + L)) + d - 1) / d);
    f.shift = L - 1;
    return f;
}
template <int STAGE>
__global__ __launch_bounds__(64) void flat(const double *__restrict__ src, double *__restrict__ jac, uint32_t total, uint32_t pps,
                                           uint32_t ppc, FlatDiv dps, FlatDiv dpc, int H, int cstride)
{
    const uint32_t at0 = blockIdx.x * 256u + threadIdx.x;
    double q[4];
    uint32_t where[4];
    bool hot[4], odd[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const uint32_t P = at0 + 64u * u;
        where[u] = P;
        hot[u] = false; odd[u] = false; q[u] = 0.0;
        if (STAGE >= 1) {
            const uint32_t Pc = P < total ? P : 0u;
            const uint32_t s = __umulhi(Pc, dps.mul) >> dps.shift, r = Pc - s * pps;
            const uint32_t col = __umulhi(r, dpc.mul) >> dpc.shift, rp = r - col * ppc;
            const int jj = (int)(col / 6u), aa = (int)(col - 6u * (uint32_t)jj);
            const int kk = (int)(rp / 6u), g = (int)(rp - 6u * (uint32_t)kk);
            const bool is_pos = g < 3;
            const int gp = is_pos ? g : g - 3;
            hot[u] = gp == (aa >> 1) && (is_pos ? (kk > jj) : (kk == jj + 1));
            odd[u] = (aa & 1) != 0;
            const int off = is_pos ? 6 * H + 6 * (jj * H - (jj * (jj - 1)) / 2 - jj - 1 + kk) + aa : 6 * jj + aa;
            const size_t idx = hot[u] ? (size_t)s * (size_t)cstride + (size_t)off : (size_t)0;
            if (STAGE == 1) q[u] = (double)idx;
            if (STAGE == 2) q[u] = src[hot[u] ? 1 : 0];
            if (STAGE == 3) q[u] = src[idx];
        }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const double v = hot[u] ? q[u] : 0.0;
        if (where[u] < total) reinterpret_cast<v2d *>(jac)[where[u]] = odd[u] ? v2d{0.0, v} : v2d{v, 0.0};
    }
}

// long-lived waves (one 64-lane workgroup per 242 KB run, like phase 3 of the fused kernel): does draining the
// wave's stores every `every` KB (s_waitcnt vmcnt(0)) or capping the waves per CU (dynamic LDS) change the rate?
__global__ __launch_bounds__(64) void fill_run_drain(double *out, int run, int every)
{
    extern __shared__ double pad[];
    v2d *o = reinterpret_cast<v2d *>(out + (size_t)blockIdx.x * run);
    const v2d z = {0.0, 0.0};
    int since = 0;
    for (int e = threadIdx.x; e < run / 2; e += 64) {
        o[e] = z;
        if (every && ++since == every) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            since = 0;
        }
    }
    if (every < 0) pad[threadIdx.x] = 0.0;
}

// Is it WHICH workgroup writes a 4 KiB block that matters?  Workgroup b writes block b + shift (wrapping inside its
// group of 8 consecutive blocks): workgroups are dealt round-robin over the 8 XCDs, so shift != 0 hands every block to
// another XCD than the one that writes it under the plain mapping.
__global__ __launch_bounds__(256) void fill_shift(double *out, size_t blocks, int shift)
{
    const size_t b = blockIdx.x;
    const size_t target = (b & ~(size_t)7) | ((b + shift) & 7);
    if (target < blocks) reinterpret_cast<v2d *>(out)[target * 256 + threadIdx.x] = v2d{0.0, 0.0};
}

static float timed(hipEvent_t e0, hipEvent_t e1)
{
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main(int argc, char **argv)
{
    const int N = argc > 1 ? atoi(argv[1]) : 65536;
    const int run = 120 * 252;
    const size_t doubles = (size_t)N * run, bytes = doubles * sizeof(double), pairs = doubles / 2;
    double *buf;
    if (hipMalloc(&buf, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto report = [&](const char *what, float ms) { printf("%-44s %.3f ms  %.0f GB/s\n", what, ms, bytes / ms / 1e6); fflush(stdout); };
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0); hipMemsetAsync(buf, 0, bytes, 0); hipEventRecord(e1);
        report("hipMemsetAsync", timed(e0, e1));
    }
    const int threads[] = {64, 128, 256, 512};
    for (int t : threads) {
#define RUN_FILL(U)                                                                                         \
    for (int rep = 0; rep < 2; ++rep) {                                                                      \
        const size_t per = (size_t)t * U;                                                                    \
        hipEventRecord(e0);                                                                                  \
        hipLaunchKernelGGL(fill_block<U>, dim3((unsigned)((pairs + per - 1) / per)), dim3(t), 0, 0, buf, pairs); \
        hipEventRecord(e1);                                                                                  \
        char name[64]; snprintf(name, sizeof name, "fill_block T=%d U=%d (%zu B per WG)", t, U, per * 16);    \
        report(name, timed(e0, e1));                                                                         \
    }
        RUN_FILL(1) RUN_FILL(2) RUN_FILL(4) RUN_FILL(8) RUN_FILL(16)
    }
    for (int t : {64, 128, 256})
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(fill_run, dim3(N), dim3(t), 0, 0, buf, run);
            hipEventRecord(e1);
            char name[64]; snprintf(name, sizeof name, "fill_run (one WG per scenario) T=%d", t);
            report(name, timed(e0, e1));
        }
    {
        double *src;
        const size_t nzb = (size_t)N * 1380 * sizeof(double);
        if (hipMalloc(&src, nzb) != hipSuccess) { printf("alloc failed\n"); return 1; }
        hipMemset(src, 0, nzb);
        const uint32_t ppc = 126, pps = 120 * 126;
        const uint32_t total = (uint32_t)((size_t)N * pps);
        const unsigned grid = (total + 255) / 256;
#define RUN_FLAT(ST)                                                                                          \
    for (int rep = 0; rep < 3; ++rep) {                                                                        \
        hipEventRecord(e0);                                                                                    \
        hipLaunchKernelGGL(flat<ST>, dim3(grid), dim3(64), 0, 0, src, buf, total, pps, ppc, flat_div(pps), flat_div(ppc), 20, 1380); \
        hipEventRecord(e1);                                                                                    \
        report("flat expand stage " #ST, timed(e0, e1));                                                       \
    }
        RUN_FLAT(0) RUN_FLAT(1) RUN_FLAT(2) RUN_FLAT(3)
        for (int shift = 0; shift < 8; ++shift)
            for (int rep = 0; rep < 2; ++rep) {
                const size_t blocks = bytes / 4096;
                hipEventRecord(e0);
                hipLaunchKernelGGL(fill_shift, dim3((unsigned)blocks), dim3(256), 0, 0, buf, blocks, shift);
                hipEventRecord(e1);
                char name[80]; snprintf(name, sizeof name, "fill_shift (block b by workgroup b-%d mod 8)", shift);
                report(name, timed(e0, e1));
            }
        for (int lds : {0})
            for (int every : {0, 2, 4, 8, 16, 64})
                for (int rep = 0; rep < 2; ++rep) {
                    hipEventRecord(e0);
                    hipLaunchKernelGGL(fill_run_drain, dim3(N), dim3(64), lds, 0, buf, run, every);
                    hipEventRecord(e1);
                    char name[80]; snprintf(name, sizeof name, "fill_run_drain lds=%d every=%d KB", lds, every);
                    report(name, timed(e0, e1));
                }
        hipFree(src);
        if (argc > 2) return 0;
    }
    // scatter after a fill of the whole buffer (nothing of it left in the caches), then in chunks right behind each chunk's fill
    for (int wide = 0; wide < 2; ++wide)
        for (int rep = 0; rep < 2; ++rep) {
            hipMemsetAsync(buf, 0, bytes, 0);
            hipEventRecord(e0);
            if (wide) hipLaunchKernelGGL(scatter<1>, dim3(N), dim3(64), 0, 0, buf, N);
            else hipLaunchKernelGGL(scatter<0>, dim3(N), dim3(64), 0, 0, buf, N);
            hipEventRecord(e1);
            printf("scatter alone (wide=%d) after a full fill: %.3f ms\n", wide, timed(e0, e1));
        }
    for (int chunk : {256, 512, 1024, 2048, 4096, 16384, 65536})
        for (int wide = 0; wide < 2; ++wide) {
            float best = 1e9;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                for (int first = 0; first < N; first += chunk) {
                    const int cnt = N - first < chunk ? N - first : chunk;
                    double *p = buf + (size_t)first * run;
                    hipMemsetAsync(p, 0, (size_t)cnt * run * sizeof(double), 0);
                    if (wide) hipLaunchKernelGGL(scatter<1>, dim3(cnt), dim3(64), 0, 0, p, cnt);
                    else hipLaunchKernelGGL(scatter<0>, dim3(cnt), dim3(64), 0, 0, p, cnt);
                }
                hipEventRecord(e1);
                const float ms = timed(e0, e1);
                if (ms < best) best = ms;
            }
            printf("fill + scatter, same stream, chunk=%6d wide=%d: %.3f ms  %.0f GB/s of dense bytes\n", chunk, wide, best, bytes / best / 1e6);
            fflush(stdout);
        }
    if (hipGetLastError() != hipSuccess) printf("HIP error\n");
    hipFree(buf);
    return 0;
}
