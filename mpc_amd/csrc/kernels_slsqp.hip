// Whole SLSQP solves resident on the device: Kraft's O(n + m) body as a one-thread-per-problem kernel.
//
// scipy's driver (_slsqp_py.py:427-456) calls the compiled core once per "round": the core returns asking for
// f and c (mode 1), for grad f and dc/dx (mode -1), or it is done.  mpc_amd/csrc/slsqp_core.c restates that core
// and, in its QP-offload form, hands the BFGS update and the QP to mpc_qp_step (kernels_qp.hip).  What was left
// on host threads - the merit function, the line search, the convergence tests - is the text of slsqp_body.inc,
// compiled here for the device (SLSQP_BODY_FULL = 0) from the very file the host core is compiled from, with
// -ffp-contract=off: the same operations in the same order, so iterates, exit modes and iteration counts equal
// the host drivers' bit for bit.  Per-problem state stays in HBM between rounds (mpc_solve_* in include/mpc_amd.h);
// each round the body kernel sorts the unfinished problems into three request lists (evaluate f and c / evaluate
// the derivatives / solve the QP) with device-side counters, and the evaluation and QP kernels of the round read
// those lists and counters from device memory: nothing but an "unfinished" count ever crosses PCIe.
#include <cmath>

#include "mpc_kernels.hpp"
#include "slsqp_core.h"

namespace {

#define MAX(a, b) ((a) > (b) ? (a) : (b))

// the BLAS-1 helpers of slsqp_core.c (same operation order; ddot's five-fold unrolling there adds its terms left to
// right, i.e. in this order)
__device__ __forceinline__ double ddot_sl(int n, const double *dx, int, const double *dy, int)
{
    double t = 0.0;
    for (int i = 0; i < n; ++i) t = t + dx[i] * dy[i];
    return t;
}
__device__ __forceinline__ void daxpy_sl(int n, double da, const double *dx, int, double *dy, int)
{
    if (n <= 0 || da == 0.0) return;
    for (int i = 0; i < n; ++i) dy[i] = dy[i] + da * dx[i];
}
__device__ __forceinline__ void dcopy_(int n, const double *dx, int, double *dy, int)
{
    for (int i = 0; i < n; ++i) dy[i] = dx[i];
}
__device__ __forceinline__ void dscal_sl(int n, double da, double *dx, int)
{
    for (int i = 0; i < n; ++i) dx[i] = da * dx[i];
}
__device__ __forceinline__ double dnrm2_(int n, const double *dx, int)
{
    double scale = 0.0, ssq = 1.0;
    if (n < 1) return 0.0;
    for (int i = 0; i < n; ++i) {
        if (dx[i] != 0.0) {
            const double ax = fabs(dx[i]);
            if (scale < ax) { ssq = 1.0 + ssq * (scale / ax) * (scale / ax); scale = ax; }
            else ssq = ssq + (ax / scale) * (ax / scale);
        }
    }
    return scale * sqrt(ssq);
}

__device__ __forceinline__ void slsqp_body_device(int m, int meq, int la, int n, double *x, const double *xl, const double *xu, double f,
                                  const double *c, const double *g, double *a, double *acc, int *iter, int *mode, const double *r,
                                  double *l, double *x0, double *mu, double *s, double *u, double *v, double *w, int *iw,
                                  slsqp_state *S)
#define SLSQP_BODY_FULL 0
#include "slsqp_body.inc"
#undef SLSQP_BODY_FULL

}  // namespace

namespace mpck {

SolveWorkspace carve_solve_workspace(void *base, int H, int rows)
{
    const size_t R = (size_t)rows, n = 6 * (size_t)H, m = 12 * ((size_t)H + 1), mg = m + 2 * n;
    const size_t nzc = 6 * (size_t)H + 3 * (size_t)H * ((size_t)H + 1), ldl = (n + 1) * n / 2 + 1;
    SolveWorkspace W;
    double *d = static_cast<double *>(base);
    auto take = [&](size_t count) { double *p = d; d += (count + 31) & ~size_t(31); return p; };   // 256-byte aligned pieces
    W.rows = rows;
    W.lb = take(n); W.ub = take(n);
    W.X = take(R * n); W.XE = take(R * n); W.X0 = take(R * n); W.S = take(R * n); W.LO = take(R * n); W.HI = take(R * n);
    W.MU = take(R * m); W.RM = take(R * mg); W.F = take(R); W.C = take(R * m); W.G = take(R * n); W.NZ = take(R * nzc);
    W.L = take(R * ldl); W.V = take(R * n); W.ACC = take(R);
    W.ST = reinterpret_cast<slsqp_state *>(take(R * (sizeof(slsqp_state) / sizeof(double))));
    int32_t *i = reinterpret_cast<int32_t *>(d);
    auto itake = [&](size_t count) { int32_t *p = i; i += (count + 63) & ~size_t(63); return p; };
    W.MODE = itake(R); W.ITER = itake(R); W.QPMODE = itake(R); W.QPFLAGS = itake(R);
    for (int h = 0; h < 2; ++h) { W.LIST_F[h] = itake(R); W.LIST_G[h] = itake(R); W.LIST_Q[h] = itake(R); }
    W.counters = itake(32);
    W.bytes = (size_t)(reinterpret_cast<char *>(i) - static_cast<char *>(base));
    return W;
}

}  // namespace mpck

namespace {

using mpck::SolveWorkspace;

static_assert(sizeof(slsqp_state) % sizeof(double) == 0, "slsqp_state rows are carved from a double array");

// x0 clipped into the box (_slsqp_py.py:268), the state of a first call (mode 0), the box itself (MPC:420-430)
__global__ void solve_begin_kernel(const mpc_params P, SolveWorkspace W, int count, const double *__restrict__ x0, double acc,
                                   int maxiter)
{
    const int n = 6 * P.horizon;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < (size_t)n) {
        double lim = P.v_lim[0];
#pragma unroll
        for (int c = 1; c < 6; ++c) lim = ((int)(idx % 6) == c) ? P.v_lim[c] : lim;
        W.lb[idx] = -lim;
        W.ub[idx] = lim;
    }
    // lists of "round -1" (slot 0): every problem is waiting for its first call of the body
    if (idx < 32) W.counters[idx] = (idx == 0 || idx == 3) ? count : 0;
    if (idx < (size_t)count) W.LIST_F[0][idx] = (int32_t)idx;
    if (idx >= (size_t)count * n) return;
    const int i = (int)(idx % n);
    double lim = P.v_lim[0];
#pragma unroll
    for (int c = 1; c < 6; ++c) lim = (i % 6 == c) ? P.v_lim[c] : lim;
    double v = x0[idx];
    v = v < -lim ? -lim : v;
    v = v > lim ? lim : v;
    W.X[idx] = v;
    W.XE[idx] = v;
    if (i == 0) {
        const size_t p = idx / n;
        W.ACC[p] = acc;
        W.ITER[p] = maxiter;          // majiter on entry = the iteration limit (_slsqp_py.py:404)
        W.MODE[p] = 0;
        W.QPMODE[p] = 0;
        W.QPFLAGS[p] = 0;
        slsqp_state z = {};
        z.offload = 1;
        W.ST[p] = z;
    }
}

// One call of the body for every unfinished problem, one wavefront per problem: the problems are those on the
// previous round's three request lists (slot `old` of the double-buffered lists and counters; they were served since),
// the vectors the body reads are staged in LDS with coalesced loads, the first lane runs the body on them, the team
// writes back what it changed.  What the problem asks for next goes to the new slot's lists.
// counters, per slot s at [8 s ..]: +0 f requests, +1 derivative requests, +2 QP requests, +3 their total (= problems
// still unfinished), +4 the QP kernel's work-queue head (the caller zeroes the new slot before the launch);
// [16..18] running totals of the three kinds.
__global__ __launch_bounds__(64) void solve_body_kernel(int H, SolveWorkspace W, int old)
{
    extern __shared__ double body_lds[];
    const int32_t *co = W.counters + 8 * old;
    int32_t *cn = W.counters + 8 * (1 - old);
    const int cF = co[0], cG = co[1], cQ = co[2];
    int k = (int)blockIdx.x, p;
    if (k < cF) p = W.LIST_F[old][k];
    else if ((k -= cF) < cG) p = W.LIST_G[old][k];
    else if ((k -= cG) < cQ) p = W.LIST_Q[old][k];
    else return;
    const int n = 6 * H, m = 12 * (H + 1), mg = m + 2 * n, lane = (int)threadIdx.x;
    double *x = body_lds, *x0 = x + n, *s = x0 + n, *u = s + n, *v = u + n, *g = v + n, *c = g + n, *mu = c + m, *r = mu + m;
    double *gx = W.X + (size_t)p * n, *gx0 = W.X0 + (size_t)p * n, *gs = W.S + (size_t)p * n, *gmu = W.MU + (size_t)p * m;
    {
        const double *gg = W.G + (size_t)p * n, *gc = W.C + (size_t)p * m, *gr = W.RM + (size_t)p * mg;
        for (int i = lane; i < n; i += 64) { x[i] = gx[i]; x0[i] = gx0[i]; s[i] = gs[i]; g[i] = gg[i]; }
        for (int i = lane; i < m; i += 64) { c[i] = gc[i]; mu[i] = gmu[i]; r[i] = gr[i]; }
    }
    __syncthreads();
    if (lane == 0) {
        int mode = W.MODE[p];
        slsqp_state S = W.ST[p];
        if (mode == MPC_SLSQP_MODE_QP) S.qp_mode = W.QPMODE[p];
        int iter = W.ITER[p];
        double acc = W.ACC[p];
        slsqp_body_device(m, 0, m, n, x, W.lb, W.ub, W.F[p], c, g, nullptr, &acc, &iter, &mode, r, nullptr, x0, mu, s, u, v, nullptr,
                          nullptr, &S);
        W.ST[p] = S;
        W.ITER[p] = iter;
        W.ACC[p] = acc;
        W.MODE[p] = mode;
        int slot = -1;
        if (mode == 1 || mode == -1) {
            const int kind = mode == 1 ? 0 : 1;
            slot = atomicAdd(&cn[kind], 1);
            (kind == 0 ? W.LIST_F : W.LIST_G)[1 - old][slot] = p;
            atomicAdd(&W.counters[16 + kind], 1);
            atomicAdd(&cn[3], 1);
        } else if (mode == MPC_SLSQP_MODE_QP) {
            W.QPFLAGS[p] = (S.qp_reset ? 1 : 0) | (S.qp_bfgs ? 2 : 0);
            slot = atomicAdd(&cn[2], 1);
            W.LIST_Q[1 - old][slot] = p;
            atomicAdd(&W.counters[18], 1);
            atomicAdd(&cn[3], 1);
        }                                                           // else: finished, MODE[p] is the exit mode
        reinterpret_cast<int *>(body_lds + 6 * n + 3 * m)[0] = mode;
    }
    __syncthreads();
    const int mode = reinterpret_cast<const int *>(body_lds + 6 * n + 3 * m)[0];
    for (int i = lane; i < n; i += 64) { gx[i] = x[i]; gx0[i] = x0[i]; gs[i] = s[i]; }
    for (int i = lane; i < m; i += 64) gmu[i] = mu[i];
    if (mode == MPC_SLSQP_MODE_QP) {
        double *gu = W.LO + (size_t)p * n, *gv = W.HI + (size_t)p * n;
        for (int i = lane; i < n; i += 64) { gu[i] = u[i]; gv[i] = v[i]; }
    } else if (mode == 1 || mode == -1) {
        double *xe = W.XE + (size_t)p * n;                          // the point to evaluate, inside the box (_clip_x_for_func)
        for (int i = lane; i < n; i += 64) {
            double t = x[i];
            t = t < W.lb[i] ? W.lb[i] : t;
            t = t > W.ub[i] ? W.ub[i] : t;
            xe[i] = t;
        }
    }
}

__global__ void solve_result_kernel(int H, SolveWorkspace W, int count, int slot, double *__restrict__ x,
                                    double *__restrict__ fun, int32_t *__restrict__ nit, int32_t *__restrict__ status,
                                    int32_t *__restrict__ stats)
{
    const int n = 6 * H;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < 4 && stats != nullptr) stats[idx] = idx < 3 ? W.counters[16 + idx] : W.counters[8 * slot + 3];
    if (idx >= (size_t)count * n) return;
    x[idx] = W.X[idx];
    if (idx % n == 0) {
        const size_t p = idx / n;
        fun[p] = W.F[p];
        nit[p] = W.ITER[p];
        status[p] = W.MODE[p];
    }
}

__global__ void copy_i32_kernel(const int32_t *__restrict__ src, int32_t *__restrict__ dst) { *dst = *src; }

}  // namespace

namespace mpck {

size_t solve_workspace_bytes(int H, int rows) { return carve_solve_workspace(nullptr, H, rows).bytes; }

hipError_t launch_solve_begin(const mpc_params &P, const SolveWorkspace &W, int count, const double *x0, double acc, int maxiter,
                              hipStream_t stream)
{
    const size_t total = (size_t)(count > 0 ? count : 1) * 6 * P.horizon;
    hipLaunchKernelGGL(solve_begin_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, P, W, count, x0, acc,
                       maxiter);
    return hipGetLastError();
}

hipError_t launch_solve_body(int H, const SolveWorkspace &W, int bound, int old_slot, hipStream_t stream)
{
    hipError_t e = hipMemsetAsync(W.counters + 8 * (1 - old_slot), 0, 8 * sizeof(int32_t), stream);
    if (e != hipSuccess || bound <= 0) return e;
    const size_t lds = sizeof(double) * (size_t)(6 * 6 * H + 3 * 12 * (H + 1) + 2);
    if (lds > 48 * 1024) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(solve_body_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(solve_body_kernel, dim3((unsigned)bound), dim3(64), lds, stream, H, W, old_slot);
    return hipGetLastError();
}

hipError_t launch_solve_result(int H, const SolveWorkspace &W, int count, int slot, double *x, double *fun, int32_t *nit,
                               int32_t *status, int32_t *stats, hipStream_t stream)
{
    const size_t total = (size_t)count * 6 * H;
    hipLaunchKernelGGL(solve_result_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, H, W, count, slot, x, fun,
                       nit, status, stats);
    return hipGetLastError();
}

hipError_t launch_copy_i32(const int32_t *src, int32_t *dst, hipStream_t stream)
{
    hipLaunchKernelGGL(copy_i32_kernel, dim3(1), dim3(1), 0, stream, src, dst);
    return hipGetLastError();
}

}  // namespace mpck
