// SLSQP's QP subproblem on the device: one workgroup per problem runs the team-parallel LSQ of
// lsq_spmd.hpp (Kraft's LSQ -> LSI -> LDP -> NNLS).  Compiled with -ffp-contract=off: every entry is
// computed with the operations and in the order of the host core (csrc/slsqp_core.c), so the step and the
// multipliers come out bit-identical to the host's.  Workgroups are persistent over the batch (problem
// p = blockIdx.x, blockIdx.x + gridDim.x, ...), each with its own slice of the workspace.
#include "lsq_spmd.hpp"
#include "mpc_kernels.hpp"
#include "slsqp_core.h"

// Teams of 256 threads, two per CU (LDS), hence two waves per SIMD and 256 VGPRs per wave: the row transform keeps
// its 2 x 12 accumulators, the row's earlier entries and a row of e in registers without spilling (with 512-thread
// teams, 128 VGPRs per wave, it spilled ~140 registers and whole solves were 8 % slower).
#ifndef MPC_QP_ROT
#define MPC_QP_ROT 0
#endif
#ifndef MPC_QP_CHAIN_ROT
#define MPC_QP_CHAIN_ROT 1       // 0: every team runs its recurrences on its first wavefront (development)
#endif
#ifndef MPC_QP_WAVES
#define MPC_QP_WAVES 2
#endif

namespace {

template <bool kLds>
struct HipTeam {
    static constexpr bool kLdsPool = kLds;       // the small pool (W.E, W.ff, ...) is LDS
    [[maybe_unused]] static constexpr bool kWave = true;       // whole wavefronts, thread t in wavefront t / 64: lsq_wave.hpp applies
    __device__ int tid() const { return (int)threadIdx.x; }
    __device__ int size() const { return (int)blockDim.x; }
    // By how many wavefronts this team's share-out of uneven work is turned (the row transform's first wavefronts carry
    // the long rows): the two teams of a CU should not both load the same SIMDs.
#if MPC_QP_ROT == 1
    __device__ int turn() const { return (blockIdx.x >= gridDim.x / 2) ? 2 : 0; }
#elif MPC_QP_ROT == 2
    __device__ int turn() const { return (blockIdx.x & 1u) ? 2 : 0; }
#else
    __device__ int turn() const { return 0; }
#endif
    // The lane number of a thread of the wavefront that runs the serial recurrences (lsq_wave.hpp), -1 for every other
    // thread.  The two teams of a CU should not run theirs on the same SIMD: the second half of the grid (the teams that
    // arrive on a CU second) uses its third wavefront.
#if MPC_QP_CHAIN_ROT
    __device__ int chain_lane() const
    {
        const int w = (blockIdx.x >= gridDim.x / 2) ? 2 : 0;
        return (int)(threadIdx.x >> 6) == w ? (int)(threadIdx.x & 63u) : -1;
    }
#else
    __device__ int chain_lane() const { return threadIdx.x < 64u ? (int)threadIdx.x : -1; }
#endif
    __device__ void sync() const { __syncthreads(); }
    // orders the LDS traffic of ONE wavefront (threads 0..63 of the team): its earlier writes are done before its later
    // reads; no other wavefront is waited for
    __device__ void wave_sync() const
    {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
};

extern __shared__ double qp_lds[];

// The team's workspace: the two matrices in its slice of global memory, the packed triangle and the vectors in
// LDS when the launch gave it room for them (lds != 0), else in the same slice.
// (A template parameter, not a run-time switch: with one origin per pointer the compiler addresses the small pool
// with LDS instructions instead of flat ones.)
// aug: the layout of Kraft's augmented problem (one variable and two rows more); the slices are sized for it.
template <bool kLds>
__device__ lsqspmd::Work team_work(double *work, int *iwork, size_t wd, size_t wi, int n, int m, int aug = 0)
{
    double *big = work + (size_t)blockIdx.x * wd;
    if constexpr (!kLds) {
        return lsqspmd::carve(big, big + lsqspmd::big_doubles(n, m, aug), iwork + (size_t)blockIdx.x * wi, n, m,
                              (int)blockDim.x, aug);
    } else {
        double *small = qp_lds;
        return lsqspmd::carve(big, small, reinterpret_cast<int *>(small + lsqspmd::small_doubles(n, m, (int)blockDim.x, aug)), n,
                              m, (int)blockDim.x, aug);
    }
}

template <bool kLds>
__global__ void __launch_bounds__(mpck::kQpThreads, MPC_QP_WAVES)
qp_kernel(int H, int count, const double *__restrict__ l, const double *__restrict__ g, const double *__restrict__ c,
          const double *__restrict__ nz, const double *__restrict__ lo, const double *__restrict__ hi, double *__restrict__ s,
          double *__restrict__ y, int32_t *__restrict__ mode, double *work, int *iwork, size_t wd, size_t wi)
{
    const int n = 6 * H, m = 12 * (H + 1), mg = m + 2 * n, nzc = 6 * H + 3 * H * (H + 1);
    const int ldl = (n + 1) * n / 2 + 1;
    const lsqspmd::Work W = team_work<kLds>(work, iwork, wd, wi, n, m);
    const HipTeam<kLds> T;
    for (int p = (int)blockIdx.x; p < count; p += (int)gridDim.x) {
        const double *lp = l + (size_t)p * ldl;
        PAR_FOR(T, e, 0, ldl) W.E[e] = lp[e];
        T.sync();
        const int md = lsqspmd::lsq(T, W, H, g + (size_t)p * n, c + (size_t)p * m, nz + (size_t)p * nzc, lo + (size_t)p * n,
                                    hi + (size_t)p * n, s + (size_t)p * n, y + (size_t)p * mg);
        if (threadIdx.x == 0) mode[p] = md;
        __syncthreads();
    }
}

// One request served by the team: the factor staged (reset, or from l), BFGS-updated and written back as the flags
// say, the QP solved, the Lagrangian gradient kept.  lo / hi / s_in / s_out: the request's own rows.  Returns the QP's
// mode.  (A function of its own: kernels_team.hip runs it inside whole solves.)
template <bool kLds, bool kAug>
__device__ __forceinline__ int qp_serve(const HipTeam<kLds> &T, const lsqspmd::Work &W, int H, int fl, double *l, const double *g,
                                        const double *c, const double *nz, double *r, double *v, const double *lo,
                                        const double *hi, const double *s_in, double *s_out)
{
    const int n = 6 * H, m = 12 * (H + 1), ldl = (n + 1) * n / 2 + 1;
    (void)m;
    QP_MARK_START(T);
    if (fl & 1) {                                            // the factor of the identity
        PAR_FOR(T, e, 0, ldl - 1) W.E[e] = 0.0;
        T.sync();
        PAR_FOR(T, i, 0, n) W.E[i * n - (i * (i - 1)) / 2] = 1.0;
    } else {
        // the factor comes from global memory: sixteen loads in flight per thread, not one
        const int per = 16 * (int)blockDim.x;
        int e0 = 0;
        for (; e0 + per <= ldl - 1; e0 += per) {
            double t[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) t[q] = l[e0 + q * (int)blockDim.x + (int)threadIdx.x];
#pragma unroll
            for (int q = 0; q < 16; ++q) W.E[e0 + q * (int)blockDim.x + (int)threadIdx.x] = t[q];
        }
        for (int e = e0 + (int)threadIdx.x; e < ldl - 1; e += (int)blockDim.x) W.E[e] = l[e];
    }
    if (fl & 2) {
        PAR_FOR(T, i, 0, n) {
            W.bb[i] = v[i];
            W.zz[i] = s_in[i];
        }
        T.sync();
        QP_MARK(T, 8);    // staging in
        lsqspmd::bfgs_update(T, W, H, W.E, g, nz, r, W.bb, W.zz, W.xs);
        QP_MARK(T, 9);    // BFGS update
    }
    T.sync();
    if (fl & 3) PAR_FOR(T, e, 0, ldl - 1) l[e] = W.E[e];
    T.sync();
    QP_MARK(T, 10);       // staging in (no update) / factor written back
    int md;
    if constexpr (kAug) {
        if (threadIdx.x == 0) {
            double rho = 100.0;                              // Kraft's l(n3), ten times larger per failed attempt
            for (int k = (fl >> 3) & 7; k > 0; --k) rho = 10.0 * rho;
            W.E[ldl - 1] = rho;
        }
        T.sync();
        md = lsqspmd::lsq<true>(T, W, H, g, c, nz, lo, hi, s_out, r, r + m);
    } else {
        // (a reset without an update leaves the factor of the identity: lsq's short forms for a diagonal e apply)
        md = lsqspmd::lsq(T, W, H, g, c, nz, lo, hi, s_out, r, nullptr, (fl & 3) == 1);
    }
    T.sync();
    if (md == 1) {
        PAR_FOR(T, i, 0, n) {
            v[i] = g[i] - lsqspmd::jac_column_dot(H, nz, r, i);
        }
    }
    return md;
}

// One major-iteration step of an offloaded solve (slsqp_core.h: slsqp_state.offload) for the problems listed
// in `index`: reset or BFGS-update the problem's LDL' factor, solve the QP, keep the new multipliers and the
// Lagrangian gradient v = g - (dc/dx)^T r for the next update.  State rows are addressed by problem, the
// request's own arrays by position in the list.  The factor is staged in the team's triangle (W.E), updated
// there, written back, and then turned in place into the QP's e.
// A request with MPC_QP_FLAG_AUG (slsqp_core.h) is Kraft's augmented problem for an inconsistent linearisation: the
// factor as it is, one more variable with the weight 100 * 10^k (k in the flags' bits 3..5); the step's extra entry
// goes to the problem's multiplier row behind the m constraint multipliers, R[p][m] (a bound row's place, unused).
// Such requests are rare and are served by an instance of their own (kAug: it takes ONLY them, the plain instance only
// the others and answers an augmented one with MPC_QP_MODE_DEFERRED, raising *aug_seen): the plain instance's code,
// registers and workspace layout stay what they are without the augmented problem (both in one kernel: +1.2 .. 1.8 %
// on whole solves that never meet one).
// kRows (device-resident solves, kernels_slsqp.hip): the request list and its length come from device memory
// (`count` is then only an upper bound), and the request's box, last step, new step and mode are the problem's own
// rows (lo, hi, s_in = s_out, mode indexed by problem instead of by position in the list; r_out unused: the
// multipliers stay in R).
template <bool kLds, bool kRows, bool kAug>
__global__ void __launch_bounds__(mpck::kQpThreads, MPC_QP_WAVES)
qp_step_kernel(int H, int count, int rows, const int32_t *__restrict__ index, const int32_t *__restrict__ count_dev,
               const int32_t *__restrict__ flags, const double *__restrict__ lo, const double *__restrict__ hi, const double *s_in,
               double *L, const double *__restrict__ G, const double *__restrict__ C, const double *__restrict__ NZ, double *R,
               double *V, double *s_out, double *__restrict__ r_out, int32_t *__restrict__ mode, int32_t *head, double *work,
               int *iwork, size_t wd, size_t wi, int32_t *aug_seen, int32_t *free_steps)
{
    const int n = 6 * H, m = 12 * (H + 1), mg = m + 2 * n, nzc = 6 * H + 3 * H * (H + 1);
    const int ldl = (n + 1) * n / 2 + 1;
    const lsqspmd::Work W = team_work<kLds>(work, iwork, wd, wi, n, m, kAug ? 1 : 0);
    const HipTeam<kLds> T;
    if (kRows) count = *count_dev;
    // kRows: the teams take requests from a shared queue (QPs differ a lot in cost: a fixed share per team would leave
    // most teams waiting for the slowest); every team leaves the loop at its first k >= count.
    for (int turn = 0;; ++turn) {
        int k = (int)blockIdx.x + turn * (int)gridDim.x;
        if (kRows) {
            if (threadIdx.x == 0) W.isc[15] = atomicAdd(head, 1);
            __syncthreads();
            k = W.isc[15];
            __syncthreads();
        }
        if (k >= count) break;
        const int pi = index[k];
        const size_t q = kRows ? (size_t)pi : (size_t)k;        // row of the per-request arrays
        if (pi < 0 || pi >= rows) {                              // never address state outside the caller's rows
            if (threadIdx.x == 0 && !kRows) mode[q] = MPC_QP_MODE_BAD_INDEX;
            continue;
        }
        const size_t p = (size_t)pi;
        double *l = L + p * ldl, *r = R + p * mg, *v = V + p * n;
        const double *g = G + p * n, *c = C + p * m, *nz = NZ + p * nzc;
        const int fl = flags[q];
        if (kAug != ((fl & MPC_QP_FLAG_AUG) != 0)) {             // the other instance's request
            if (!kAug && threadIdx.x == 0) {
                mode[q] = MPC_QP_MODE_DEFERRED;
                if (aug_seen) *aug_seen = 1;
            }
            continue;
        }
        if (threadIdx.x == 0) W.isc[5] = 0;
        const int md = qp_serve<kLds, kAug>(T, W, H, fl, l, g, c, nz, r, v, lo + q * n, hi + q * n, s_in + q * n, s_out + q * n);
        if (free_steps && threadIdx.x == 0 && W.isc[5] == 1) atomicAdd(free_steps, 1);     // (lsq: the unconstrained step sufficed)
        if (md == 1 && !kRows) PAR_FOR(T, i, 0, m) r_out[q * m + i] = r[i];
        if (threadIdx.x == 0) mode[q] = md;
        __syncthreads();
        QP_MARK_START(T);
        QP_MARK(T, 11);       // (nothing: calibration of the mark itself)
    }
}

#if defined(MPC_QP_PROFILE)
}  // namespace
extern "C" int mpc_debug_qp_profile(unsigned long long *out32, int reset)
{
    hipError_t e = hipMemcpyFromSymbol(out32, HIP_SYMBOL(mpc_qp_prof), 32 * sizeof(unsigned long long));
    if (e == hipSuccess && reset) {
        unsigned long long z[32] = {0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(mpc_qp_prof), z, sizeof(z));
    }
    return e == hipSuccess ? 0 : -2;
}
namespace {
#endif

// LDS of one team, in bytes, when its small pool goes there (0: it stays in global memory).
size_t lds_bytes(int H, int aug = 0)
{
    const int n = 6 * H, m = 12 * (H + 1);
    const size_t b = lsqspmd::small_doubles(n, m, mpck::kQpThreads, aug) * sizeof(double) +
                     lsqspmd::work_ints(n, m, mpck::kQpThreads, aug) * sizeof(int);
    return b <= mpck::kQpLdsLimit ? b : 0;
}

}  // namespace

#ifndef MPC_TEAM_UNIT       // (kernels_team.hip includes this file for its device code only)
namespace mpck {

int qp_teams_per_cu(int H)
{
#ifdef MPC_QP_TEAMS_PER_CU          // development: measure a team without a neighbour on its CU
    return MPC_QP_TEAMS_PER_CU;
#endif
    const size_t b = lds_bytes(H);
    if (!b) return 4;
    const int t = (int)(kQpLdsLimit / b);
    return t < 1 ? 1 : (t > 4 ? 4 : t);
}

namespace {
template <bool kRows, bool kAug>
hipError_t launch_qp_step_any(int H, int count, int teams, int rows, const int32_t *index, const int32_t *count_dev,
                              const int32_t *flags, const double *lo, const double *hi, const double *s_in, double *L, const double *G,
                              const double *C, const double *NZ, double *R, double *V, double *s_out, double *r_out, int32_t *mode,
                              int32_t *head, double *work, int *iwork, int32_t *aug_seen, int32_t *free_steps,
                              hipStream_t stream)
{
    const size_t lds = lds_bytes(H, kAug ? 1 : 0);
    if (lds) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(qp_step_kernel<true, kRows, kAug>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((qp_step_kernel<true, kRows, kAug>), dim3((unsigned)teams), dim3(kQpThreads), lds, stream, H, count, rows,
                           index, count_dev, flags, lo, hi, s_in, L, G, C, NZ, R, V, s_out, r_out, mode, head, work, iwork,
                           qp_work_doubles(H), qp_work_ints(H), aug_seen, free_steps);
    } else {
        hipLaunchKernelGGL((qp_step_kernel<false, kRows, kAug>), dim3((unsigned)teams), dim3(kQpThreads), 0, stream, H, count, rows,
                           index, count_dev, flags, lo, hi, s_in, L, G, C, NZ, R, V, s_out, r_out, mode, head, work, iwork,
                           qp_work_doubles(H), qp_work_ints(H), aug_seen, free_steps);
    }
    return hipGetLastError();
}
}  // namespace

hipError_t launch_qp_step(int H, int count, int teams, int rows, const int32_t *index, const int32_t *flags, const double *lo,
                          const double *hi, const double *s_in, double *L, const double *G, const double *C, const double *NZ,
                          double *R, double *V, double *s_out, double *r_out, int32_t *mode, double *work, int *iwork,
                          hipStream_t stream)
{
    return launch_qp_step_any<false, false>(H, count, teams, rows, index, nullptr, flags, lo, hi, s_in, L, G, C, NZ, R, V, s_out,
                                            r_out, mode, nullptr, work, iwork, nullptr, nullptr, stream);
}

hipError_t launch_qp_step_rows(int H, int teams, int rows, const RowList &list, const int32_t *flags, const double *lo,
                               const double *hi, double *S, double *L, const double *G, const double *C, const double *NZ,
                               double *R, double *V, int32_t *mode, double *work, int *iwork, int32_t *aug_seen, bool augmented,
                               int32_t *free_steps, hipStream_t stream)
{
    if (list.bound <= 0) return hipSuccess;
    if (list.head == nullptr) return hipErrorInvalidValue;
    const int t = teams < list.bound ? teams : list.bound;
    if (augmented)
        return launch_qp_step_any<true, true>(H, list.bound, t, rows, list.list, list.count, flags, lo, hi, S, L, G, C, NZ, R, V, S,
                                              nullptr, mode, list.head, work, iwork, nullptr, nullptr, stream);
    return launch_qp_step_any<true, false>(H, list.bound, t, rows, list.list, list.count, flags, lo, hi, S, L, G, C, NZ, R, V, S,
                                           nullptr, mode, list.head, work, iwork, aug_seen, free_steps, stream);
}

// (sized for the augmented problem's layout, the larger of the two)
size_t qp_work_doubles(int H) { return lsqspmd::work_doubles(6 * H, 12 * (H + 1), kQpThreads, 1); }
size_t qp_work_ints(int H) { return lsqspmd::work_ints(6 * H, 12 * (H + 1), kQpThreads, 1); }

hipError_t launch_qp(int H, int count, int teams, const double *l, const double *g, const double *c, const double *nz,
                     const double *lo, const double *hi, double *s, double *y, int32_t *mode, double *work, int *iwork,
                     hipStream_t stream)
{
    const size_t lds = lds_bytes(H);
    if (lds) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(qp_kernel<true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(qp_kernel<true>, dim3((unsigned)teams), dim3(kQpThreads), lds, stream, H, count, l, g, c, nz, lo, hi, s,
                           y, mode, work, iwork, qp_work_doubles(H), qp_work_ints(H));
    } else {
        hipLaunchKernelGGL(qp_kernel<false>, dim3((unsigned)teams), dim3(kQpThreads), 0, stream, H, count, l, g, c, nz, lo, hi, s,
                           y, mode, work, iwork, qp_work_doubles(H), qp_work_ints(H));
    }
    return hipGetLastError();
}

}  // namespace mpck
#endif  // MPC_TEAM_UNIT
