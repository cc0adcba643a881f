// Whole SLSQP solves resident on the device: Kraft's O(n + m) body as a one-wavefront-per-problem kernel.
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
#include "lsq_wave.hpp"
#include "slsqp_core.h"

namespace {

#define MAX(a, b) ((a) > (b) ? (a) : (b))

#if defined(__HIP_DEVICE_COMPILE__)      // (lsq_wave.hpp exists in the device pass only)
// The helpers of slsqp_core.c for ONE WAVEFRONT that runs the body on one problem: all 64 lanes execute the body's
// text with the same scalars; the vectors live in LDS, entry i is the business of lane i % 64, and every ordered sum
// (ddot's five-fold unrolling on the host adds its terms left to right: the same chain) is one chain of additions in
// the order of the serial text, run by lsq_wave.hpp's ordered_sum_from at the FP64 pipeline's latency instead of one
// LDS round trip per term: the terms are formed by the lanes, 128 at a time.
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }
__device__ __forceinline__ void wave_sync() { lsqwave::wave_sync(); }

// t + term(0) + term(1) + ... + term(count - 1), in that order.  A block of fewer than 128 terms sits at the END of
// the 128 slots and the chain is entered at its first term: nothing is added that the serial text does not add.
template <class Term>
__device__ __forceinline__ double wave_ordered_sum(double t, int count, Term term)
{
    const int c = lane_id() & 15;
    wave_sync();                                   // the vectors the terms are formed from were written by other lanes
    for (int base = 0; base < count; base += 128) {
        const int k = count - base < 128 ? count - base : 128, pad = 128 - k;
        double P[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int slot = 16 * r + c;
            P[r] = slot >= pad ? term(base + slot - pad) : 0.0;
        }
        t = lsqwave::ordered_sum_from(t, P, pad);
    }
    return t;
}

__device__ __forceinline__ double ddot_sl(int n, const double *dx, int, const double *dy, int)
{
    return wave_ordered_sum(0.0, n, [&](int i) { return dx[i] * dy[i]; });
}
__device__ __forceinline__ void daxpy_sl(int n, double da, const double *dx, int, double *dy, int)
{
    if (n <= 0 || da == 0.0) return;
    for (int i = lane_id(); i < n; i += 64) dy[i] = dy[i] + da * dx[i];
    wave_sync();
}
__device__ __forceinline__ void dcopy_(int n, const double *dx, int, double *dy, int)
{
    for (int i = lane_id(); i < n; i += 64) dy[i] = dx[i];
    wave_sync();
}
__device__ __forceinline__ void dscal_sl(int n, double da, double *dx, int)
{
    for (int i = lane_id(); i < n; i += 64) dx[i] = da * dx[i];
    wave_sync();
}
// the scaled sum of squares is a recurrence with a branch per term: every lane runs it on the same entries
__device__ __forceinline__ double dnrm2_(int n, const double *dx, int)
{
    double scale = 0.0, ssq = 1.0;
    if (n < 1) return 0.0;
    wave_sync();
    for (int i = 0; i < n; ++i) {
        if (dx[i] != 0.0) {
            const double ax = fabs(dx[i]);
            if (scale < ax) { ssq = 1.0 + ssq * (scale / ax) * (scale / ax); scale = ax; }
            else ssq = ssq + (ax / scale) * (ax / scale);
        }
    }
    return scale * sqrt(ssq);
}
// the loops of the body over the constraints (slsqp_core.c has the serial forms)
__device__ __forceinline__ void sl_fill(int n, double *x, double v)
{
    for (int i = lane_id(); i < n; i += 64) x[i] = v;
    wave_sync();
}
__device__ __forceinline__ double sl_sum_infeas(int m, int meq, const double *c, double t)
{
    return wave_ordered_sum(t, m, [&](int j) {
        const double h = (j < meq) ? c[j] : 0.0;
        return MAX(-c[j], h);
    });
}
__device__ __forceinline__ double sl_sum_mu_infeas(int m, int meq, const double *mu, const double *c, double t)
{
    return wave_ordered_sum(t, m, [&](int j) {
        const double h = (j < meq) ? c[j] : 0.0;
        return mu[j] * MAX(-c[j], h);
    });
}
__device__ __forceinline__ double sl_sum_abs_prod(int m, const double *r, const double *c, double t)
{
    return wave_ordered_sum(t, m, [&](int j) { return fabs(r[j]) * fabs(c[j]); });
}
__device__ __forceinline__ void sl_mu_update(int m, const double *r, double *mu)
{
    for (int j = lane_id(); j < m; j += 64) {
        const double h = fabs(r[j]);
        mu[j] = MAX(h, (mu[j] + h) / 2.0);
    }
    wave_sync();
}

// (no __restrict__ here: the lanes of the wavefront read what other lanes wrote)
__device__ __forceinline__ void slsqp_body_device(int m, int meq, int la, int n, double *x, const double *xl, const double *xu,
                                                  double f, const double *c, const double *g, double *a, double *acc, int *iter,
                                                  int *mode, const double *r, double *l, double *x0, double *mu, double *s,
                                                  double *u, double *v, double *w, int *iw, slsqp_state *S)
#define SLSQP_BODY_FULL 0
#include "slsqp_body.inc"
#undef SLSQP_BODY_FULL
#endif  // __HIP_DEVICE_COMPILE__

}  // namespace

#ifndef MPC_TEAM_UNIT       // (kernels_team.hip includes this file for its device code only)
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
#endif  // MPC_TEAM_UNIT

namespace {

using mpck::SolveWorkspace;

static_assert(sizeof(slsqp_state) % sizeof(double) == 0, "slsqp_state rows are carved from a double array");

constexpr int kTotals = 24, kOverflow = 27;      // places in W.counters beyond the three slots of eight

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
    // lists of "round -1" (counter slot 0): every problem is waiting for its first call of the body
    // (the seed list is not a round of f requests: the running total starts at -count, see solve_body_kernel)
    if (idx < 32) W.counters[idx] = (idx == 0 || idx == 3) ? count : (idx == kTotals ? -count : 0);
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
        W.ITER[p] = maxiter - 1;      // majiter on entry = the iteration limit, maxiter - 1 (_slsqp_py.py:245,392)
        W.MODE[p] = 0;
        W.QPMODE[p] = 0;
        W.QPFLAGS[p] = 0;
        slsqp_state z = {};
        z.offload = 2;                 // the QP kernel takes the augmented problem as well
        W.ST[p] = z;
    }
}

// One call of the body for every unfinished problem, one wavefront per problem: the problems are those on the
// previous round's three request lists (slot `old` of the double-buffered lists and counters; they were served since),
// the vectors the body reads are staged in LDS with coalesced loads, the wavefront runs the body on them, the team
// writes back what it changed.  What the problem asks for next goes to the new slot's lists.
// The request lists alternate between two buffers (round r reads buffer r % 2 and writes the other); their counters
// rotate through THREE slots, so that a round can clear the slot of the round after it and no memset sits between
// the rounds: round r reads slot r % 3, fills slot (r + 1) % 3 and zeroes slot (r + 2) % 3, which nothing has read
// since round r - 1's body.
// counters, per slot s at [8 s ..]: +0 f requests, +1 derivative requests, +2 QP requests, +3 their total (= problems
// still unfinished, or -1 if a round met more requests than its grid was sized for; written by solve_count_kernel),
// +4 the QP kernel's work-queue head, +5 that of its instance for augmented problems; [kTotals ..] running totals of the three kinds over every round up to and
// including the old slot's (the first workgroup adds the old slot's lengths), [kOverflow] the sticky flag "a round's
// `bound` was smaller than its lists".  One atomic per kind and workgroup (every problem of a lock-step round would
// increment the same word, and L2 serialises them).
constexpr int kBodyWaves = 4;      // problems per workgroup of solve_body_kernel, one wavefront each (fewer if LDS is short)

// ---- one call of the body for one problem, by one wavefront: the pieces ---------------------------------------------
// (solve_body_kernel strings them together for the problems of a lock-step round; kernels_team.hip for one problem after
//  the other inside a workgroup that runs whole solves)
struct BodyVectors { double *x, *x0, *s, *u, *v, *g, *c, *mu, *r; };       // LDS: 6 n + 3 m doubles
__device__ __forceinline__ BodyVectors body_vectors(double *base, int n, int m)
{
    BodyVectors B;
    B.x = base; B.x0 = B.x + n; B.s = B.x0 + n; B.u = B.s + n; B.v = B.u + n; B.g = B.v + n; B.c = B.g + n; B.mu = B.c + m;
    B.r = B.mu + m;
    return B;
}
struct BodyScalars { int mode, iter, qp_mode; double acc, fval, qp_delta; };
// the problem's scalars are requested with the vectors, not after the barrier where they are used (a second and, for
// the QP's answer, a third trip to memory on a kernel that is little else than latency)
__device__ __forceinline__ BodyScalars body_scalars(const SolveWorkspace &W, int p, int mg, int m)
{
    BodyScalars Z;
    Z.mode = W.MODE[p];
    Z.iter = W.ITER[p];
    Z.qp_mode = W.QPMODE[p];
    Z.acc = W.ACC[p];
    Z.fval = W.F[p];
    Z.qp_delta = W.RM[(size_t)p * mg + m];                          // an augmented problem's extra entry of the step
    return Z;
}
// every load of up to 256 entries of the seven vectors is issued before the first LDS store: one trip to memory per
// pass (a single pass at H = 20) instead of one per vector and stride
__device__ __forceinline__ void body_stage(const SolveWorkspace &W, size_t pp, int n, int m, int mg, int lane, const BodyVectors &B)
{
    const double *gx = W.X + pp * n, *gx0 = W.X0 + pp * n, *gs = W.S + pp * n, *gmu = W.MU + pp * m;
    const double *gg = W.G + pp * n, *gc = W.C + pp * m, *gr = W.RM + pp * mg;
    for (int base = 0; base < m; base += 256) {
        double rx[4], rx0[4], rs[4], rg[4], rc[4], rmu[4], rr[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = base + 64 * q + lane;
            const bool in_n = i < n, in_m = i < m;
            rx[q] = in_n ? gx[i] : 0.0;
            rx0[q] = in_n ? gx0[i] : 0.0;
            rs[q] = in_n ? gs[i] : 0.0;
            rg[q] = in_n ? gg[i] : 0.0;
            rc[q] = in_m ? gc[i] : 0.0;
            rmu[q] = in_m ? gmu[i] : 0.0;
            rr[q] = in_m ? gr[i] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = base + 64 * q + lane;
            if (i < n) { B.x[i] = rx[q]; B.x0[i] = rx0[q]; B.s[i] = rs[q]; B.g[i] = rg[q]; }
            if (i < m) { B.c[i] = rc[q]; B.mu[i] = rmu[q]; B.r[i] = rr[q]; }
        }
    }
}
// The whole wavefront runs the body (same scalars in every lane, the vectors shared out); its first lane files the
// scalars away.  Returns what the problem asks for next: 0 f and c, 1 derivatives, 2 its QP, -1 nothing (finished);
// *box_fresh: the body formed the QP's box in u, v (not so when it asks again, augmented or deferred).
// (A problem whose augmented QP met the plain QP kernel in a round without the augmented one - mpc_solve_rounds
//  launches that only once such a request has been seen - has not been served: it asks again, as it stands.)
__device__ __forceinline__ int body_run(const SolveWorkspace &W, int p, int n, int m, int lane, BodyScalars Z, const BodyVectors &B,
                                        bool *box_fresh)
{
    *box_fresh = false;
    if (Z.mode == MPC_SLSQP_MODE_QP && Z.qp_mode == MPC_QP_MODE_DEFERRED) return 2;
    slsqp_state S = W.ST[p];
    if (Z.mode == MPC_SLSQP_MODE_QP) {
        S.qp_mode = Z.qp_mode;
        S.qp_delta = Z.qp_delta;
    }
    int mode = Z.mode, iter = Z.iter;
    double acc = Z.acc;
#if defined(__HIP_DEVICE_COMPILE__)
    slsqp_body_device(m, 0, m, n, B.x, W.lb, W.ub, Z.fval, B.c, B.g, nullptr, &acc, &iter, &mode, B.r, nullptr, B.x0, B.mu, B.s,
                      B.u, B.v, nullptr, nullptr, &S);
#endif
    int kind = -1;
    if (mode == 1) kind = 0;
    else if (mode == -1) kind = 1;
    else if (mode == MPC_SLSQP_MODE_QP) kind = 2;
    *box_fresh = kind == 2 && !S.qp_aug;
    if (lane == 0) {
        W.ST[p] = S;
        W.ITER[p] = iter;
        W.ACC[p] = acc;
        W.MODE[p] = mode;                                           // not 1, -1 or QP: finished, this is the exit mode
        if (kind == 2) W.QPFLAGS[p] = MPC_QP_FLAGS_OF(&S);
    }
    return kind;
}
// what the body changed goes back to the problem's rows; the QP's box or the clipped evaluation point with it
__device__ __forceinline__ void body_store(const SolveWorkspace &W, size_t pp, int n, int m, int lane, const BodyVectors &B, int kind,
                                           bool box_fresh)
{
    double *gx = W.X + pp * n, *gx0 = W.X0 + pp * n, *gs = W.S + pp * n, *gmu = W.MU + pp * m;
    for (int i = lane; i < n; i += 64) { gx[i] = B.x[i]; gx0[i] = B.x0[i]; gs[i] = B.s[i]; }
    for (int i = lane; i < m; i += 64) gmu[i] = B.mu[i];
    if (kind == 2) {
        if (box_fresh) {
            double *gu = W.LO + pp * n, *gv = W.HI + pp * n;
            for (int i = lane; i < n; i += 64) { gu[i] = B.u[i]; gv[i] = B.v[i]; }
        }
    } else if (kind >= 0) {
        double *xe = W.XE + pp * n;                                 // the point to evaluate, inside the box (_clip_x_for_func)
        for (int i = lane; i < n; i += 64) {
            double t = B.x[i];
            t = t < W.lb[i] ? W.lb[i] : t;
            t = t > W.ub[i] ? W.ub[i] : t;
            xe[i] = t;
        }
    }
}

// (three workgroups per CU is what their LDS allows at H = 20: the registers are held to that, 168 per lane)
__global__ __launch_bounds__(64 * kBodyWaves, 3) void solve_body_kernel(int H, SolveWorkspace W, int old, int cs_old, int cs_new,
                                                                        int cs_next)
{
    extern __shared__ double body_lds[];
    __shared__ int ask[kBodyWaves];          // what each wavefront's problem asks for: 0 f, 1 derivatives, 2 QP, -1 nothing
    __shared__ int first[3];                 // first list position of this workgroup's requests, per kind
    const int32_t *co = W.counters + 8 * cs_old;
    int32_t *cn = W.counters + 8 * cs_new;
    const int cF = co[0], cG = co[1], cQ = co[2];
    if (blockIdx.x == 0) {
        if (threadIdx.x < 3) W.counters[kTotals + threadIdx.x] += co[threadIdx.x];   // the round just served
        if (threadIdx.x < 8) W.counters[8 * cs_next + threadIdx.x] = 0;               // the next round's slot
        if (threadIdx.x == 8 && (long long)cF + cG + cQ > (long long)gridDim.x * (blockDim.x >> 6)) W.counters[kOverflow] = 1;
    }
    const int n = 6 * H, m = 12 * (H + 1), mg = m + 2 * n, lane = (int)(threadIdx.x & 63u), wave = (int)(threadIdx.x >> 6);
    const int waves = (int)(blockDim.x >> 6);
    int k = (int)blockIdx.x * waves + wave, p = -1;
    if (k < cF) p = W.LIST_F[old][k];
    else if ((k -= cF) < cG) p = W.LIST_G[old][k];
    else if ((k -= cG) < cQ) p = W.LIST_Q[old][k];
    p = __builtin_amdgcn_readfirstlane(p);   // one problem per wavefront: said so, its scalars then live in scalar registers
    const bool live = p >= 0;                // a wavefront beyond the lists only takes part in the barriers
    const BodyVectors B = body_vectors(body_lds + wave * (size_t)(6 * n + 3 * m), n, m);
    const size_t pp = live ? (size_t)p : 0;
    BodyScalars Z = {};
    if (live) {
        Z = body_scalars(W, p, mg, m);
        body_stage(W, pp, n, m, mg, lane, B);
    }
    __syncthreads();
    bool box_fresh = false;
    {
        int kind = -1;
        if (live) kind = body_run(W, p, n, m, lane, Z, B, &box_fresh);
        if (lane == 0) ask[wave] = kind;
    }
    __syncthreads();
    // one atomic per kind and WORKGROUP reserves the list positions of its requests (every problem of a lock-step round
    // increments the same word, and L2 serialises them)
    if (threadIdx.x < 3) {
        int count = 0;
        for (int w = 0; w < waves; ++w) count += ask[w] == (int)threadIdx.x;
        first[threadIdx.x] = count ? atomicAdd(&cn[threadIdx.x], count) : 0;
    }
    __syncthreads();
    const int kind = ask[wave];
    if (!live) return;
    if (lane == 0 && kind >= 0) {
        int slot = first[kind];
        for (int w = 0; w < wave; ++w) slot += ask[w] == kind;
        (kind == 0 ? W.LIST_F : kind == 1 ? W.LIST_G : W.LIST_Q)[1 - old][slot] = p;
    }
    body_store(W, pp, n, m, lane, B, kind, box_fresh);
}

__global__ void solve_result_kernel(int H, SolveWorkspace W, int count, int slot, double *__restrict__ x,
                                    double *__restrict__ fun, int32_t *__restrict__ nit, int32_t *__restrict__ status,
                                    int32_t *__restrict__ stats)
{
    const int n = 6 * H;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < 5 && stats != nullptr) {     // the latest slot's requests have been served but not yet added to the totals
        const int32_t *c = W.counters + 8 * slot;
        stats[idx] = idx < 3 ? W.counters[kTotals + idx] + c[idx]
                             : (idx == 4 ? W.counters[mpck::kSolveFreeSteps] : (W.counters[kOverflow] ? -1 : c[0] + c[1] + c[2]));
    }
    if (idx >= (size_t)count * n) return;
    x[idx] = W.X[idx];
    if (idx % n == 0) {
        const size_t p = idx / n;
        fun[p] = W.F[p];
        nit[p] = W.ITER[p];
        status[p] = W.MODE[p];
    }
}

// unfinished problems of a round = the lengths of its three request lists; -1 once a round was launched with a grid
// smaller than its lists (problems were left out of it: the solve is void)
__global__ void solve_count_kernel(int32_t *__restrict__ c, const int32_t *__restrict__ overflow)
{
    c[3] = *overflow ? -1 : c[0] + c[1] + c[2];
}

}  // namespace

#ifndef MPC_TEAM_UNIT       // (kernels_team.hip includes this file for its device code only)
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

hipError_t launch_solve_body(int H, const SolveWorkspace &W, int bound, int round, hipStream_t stream)
{
    if (bound <= 0) return hipSuccess;
    hipError_t e = hipSuccess;
    const size_t per = sizeof(double) * (size_t)(6 * 6 * H + 3 * 12 * (H + 1));
    int waves = (int)((size_t)150 * 1024 / per);
    waves = waves < 1 ? 1 : (waves > kBodyWaves ? kBodyWaves : waves);
    const size_t lds = per * (size_t)waves;
    if (lds > 48 * 1024) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(solve_body_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(solve_body_kernel, dim3((unsigned)((bound + waves - 1) / waves)), dim3(64 * waves), lds, stream, H, W,
                       round & 1, round % 3, (round + 1) % 3, (round + 2) % 3);
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

namespace {
// test hook: one wavefront per case, slot s = 16 r + c of the case in register r of the lanes with lane % 16 == c
__global__ __launch_bounds__(64) void ordered_sum_probe_kernel(int count, const double *__restrict__ terms,
                                                               const double *__restrict__ t0, const int32_t *__restrict__ first,
                                                               double *__restrict__ out)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const int i = (int)blockIdx.x, c = (int)(threadIdx.x & 15u);
    if (i >= count) return;
    double P[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) P[r] = terms[(size_t)i * 128 + 16 * r + c];
    int f = first[i];
    f = f < 0 ? 0 : (f > 128 ? 128 : f);
    const double t = lsqwave::ordered_sum_from(t0[i], P, __builtin_amdgcn_readfirstlane(f));
    if (threadIdx.x == 63) out[i] = t;           // every lane holds the sum: the last one says so
#endif
}
}  // namespace

hipError_t launch_ordered_sum_probe(int count, const double *terms, const double *t0, const int32_t *first, double *out,
                                    hipStream_t stream)
{
    hipLaunchKernelGGL(ordered_sum_probe_kernel, dim3((unsigned)count), dim3(64), 0, stream, count, terms, t0, first, out);
    return hipGetLastError();
}

hipError_t launch_solve_count(const SolveWorkspace &W, int slot, hipStream_t stream)
{
    hipLaunchKernelGGL(solve_count_kernel, dim3(1), dim3(1), 0, stream, W.counters + 8 * slot, W.counters + kOverflow);
    return hipGetLastError();
}

}  // namespace mpck
#endif  // MPC_TEAM_UNIT
