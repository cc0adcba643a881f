// Whole SLSQP solves, one workgroup per problem at a time, no lock step.
//
// mpc_solve_rounds (kernels_slsqp.hip) advances every unfinished problem by one call of Kraft's body per round and
// serves the requests of a round with one evaluation / QP kernel each: right for batches that fill the chip, but a
// round costs at least one QP's latency (0.25-0.4 ms) however few problems ask for one, and problems drift out of
// step, so a small batch pays rounds x latency (4,096 problems: 106 ms for 60 ms of QP work; one problem: ~100 rounds
// of five launches).  Here a persistent team of 256 threads takes a problem from a queue and runs it to its end:
//     body (first wavefront)  ->  f, c  |  grad f, structured dc/dx  |  BFGS update + QP   ->  body  -> ...
// with the very device code of the lock-step kernels (body_* of kernels_slsqp.hip, prefix_body of kernels_prefix.hip,
// qp_serve of kernels_qp.hip: this unit includes those files for their device code), on the same per-problem rows of
// the caller's workspace, so iterates, exit modes and counts are the lock-step path's, bit for bit.  Between two QPs
// the team's LDS is free: the body's vectors sit at its start, the evaluation's scenario behind them.
// Kraft's augmented problem (inconsistent linearisation) is served by a function that is not inlined (see
// kernels_qp.hip on what its text costs the plain problem's registers when it is).
#define MPC_TEAM_UNIT 1
#include "kernels_prefix.hip"
#include "kernels_qp.hip"
#include "kernels_slsqp.hip"

namespace {

// The QP behind a call, the plain problem as well as the augmented one: a function that is not inlined gets registers of
// its own, so the QP runs with the allocation it has in qp_step_kernel whatever the rest of this kernel needs (inlined
// beside the evaluations and the body it spilled 325 registers and a QP took a quarter longer).  The workspace is
// carved again inside (LDS stays LDS); the problem's rows arrive as generic pointers and are said to be global again.
#if defined(__HIP_DEVICE_COMPILE__)
template <bool kAug>
__device__ __attribute__((noinline)) int qp_serve_call(double *work, int *iwork, size_t wd, size_t wi, int H, int fl, double *l,
                                                       const double *g, const double *c, const double *nz, double *r, double *v,
                                                       const double *lo, const double *hi, double *s, int32_t *free_steps)
{
    using lsqspmd::global_again;
    H = __builtin_amdgcn_readfirstlane(H);
    fl = __builtin_amdgcn_readfirstlane(fl);
    wd = (size_t)lsqspmd::scalar_bits((const void *)wd);
    wi = (size_t)lsqspmd::scalar_bits((const void *)wi);
    work = global_again(work);
    iwork = (int *)global_again((const double *)iwork);
    const lsqspmd::Work W = team_work<true>(work, iwork, wd, wi, 6 * H, 12 * (H + 1), kAug ? 1 : 0);
    s = global_again(s);
    if (threadIdx.x == 0) W.isc[5] = 0;
    const int md = qp_serve<true, kAug>(HipTeam<true>(), W, H, fl, global_again(l), global_again(g), global_again(c),
                                        global_again(nz), global_again(r), global_again(v), global_again(lo), global_again(hi), s, s);
    if (threadIdx.x == 0 && W.isc[5] == 1) atomicAdd(free_steps, 1);
    return md;
}
#endif

constexpr int kTeamLeft = 28;      // place in W.counters: problems a team gave up on (the bound on calls of the body)

template <bool SAME_DT>
__global__ void __launch_bounds__(mpck::kQpThreads, MPC_QP_WAVES)
solve_team_kernel(const mpc_params P, PrefixArgs A, SolveWorkspace W, int count, int max_calls, double *work, int *iwork, size_t wd,
                  size_t wi)
{
#if defined(__HIP_DEVICE_COMPILE__)          // (the wavefront helpers exist in the device pass only)
    __shared__ int note[4];                  // [0] the problem taken from the queue, [1] what it asks for
    const int H = P.horizon, n = 6 * H, m = 12 * (H + 1), mg = m + 2 * n, nzc = 6 * H + 3 * H * (H + 1);
    const int ldl = (n + 1) * n / 2 + 1;
    const int tid = (int)threadIdx.x, lane = tid & 63;
    const BodyVectors B = body_vectors(qp_lds, n, m);
    double *eval_lds = qp_lds + ((6 * n + 3 * m + 1) & ~1);        // 16-byte aligned, behind the body's vectors
    int32_t *head = W.counters + 4;                                // the queue: slot 0's QP work-queue head, unused here
    int n_f = 0, n_g = 0, n_q = 0;                                 // thread 0: requests served, for the statistics
    for (;;) {
        if (tid == 0) note[0] = atomicAdd(head, 1);
        __syncthreads();
        const int p = note[0];
        __syncthreads();
        if (p >= count) break;
        const size_t pp = (size_t)p;
        bool finished = false;
        for (int call = 0; call < max_calls; ++call) {
            if (tid < 64) {                                        // the first wavefront runs the body
                const BodyScalars Z = body_scalars(W, p, mg, m);
                body_stage(W, pp, n, m, mg, lane, B);
                wave_sync();
                bool box_fresh = false;
                const int kind = body_run(W, p, n, m, lane, Z, B, &box_fresh);
                wave_sync();
                body_store(W, pp, n, m, lane, B, kind, box_fresh);
                if (lane == 0) note[1] = kind;
            }
            __syncthreads();                                       // its stores to the problem's rows are the team's to read
            const int kind = note[1];
            if (kind < 0) { finished = true; break; }
            if (kind == 0) {                                       // f and c at the clipped point (SLSQP:434-436)
                PrefixArgs E = A;
                E.x = W.XE; E.cost = W.F; E.grad = nullptr; E.con = W.C; E.jac = nullptr; E.compact = nullptr;
                E.single_row = p;
                prefix_body<kEval, SAME_DT, true>(P, E, eval_lds, tid, (int)blockDim.x, 0);
                n_f += 1;
            } else if (kind == 1) {                                // grad f and the structured dc/dx (SLSQP:438-440)
                PrefixArgs E = A;
                E.x = W.XE; E.cost = W.F; E.grad = W.G; E.con = nullptr; E.jac = W.NZ; E.compact = W.NZ; E.cstride = nzc;
                E.tmajor = 0; E.single_row = p;
                prefix_body<kCompact, SAME_DT, true>(P, E, eval_lds, tid, (int)blockDim.x, 0);
                n_g += 1;
            } else {                                               // BFGS update + QP
                const int fl = W.QPFLAGS[p];
                double *l = W.L + pp * ldl, *r = W.RM + pp * mg, *v = W.V + pp * n, *s = W.S + pp * n;
                const double *g = W.G + pp * n, *c = W.C + pp * m, *nz = W.NZ + pp * nzc;
                int md;
                __syncthreads();                                   // (everyone has read the flags before the QP's LDS is written)
                if (fl & MPC_QP_FLAG_AUG) md = qp_serve_call<true>(work, iwork, wd, wi, H, fl, l, g, c, nz, r, v, W.LO + pp * n,
                                                                   W.HI + pp * n, s, W.counters + mpck::kSolveFreeSteps);
                else md = qp_serve_call<false>(work, iwork, wd, wi, H, fl, l, g, c, nz, r, v, W.LO + pp * n, W.HI + pp * n, s,
                                               W.counters + mpck::kSolveFreeSteps);
                if (tid == 0) W.QPMODE[p] = md;
                n_q += 1;
            }
            __syncthreads();                                       // the answer is in the problem's rows before the body reads them
        }
        if (!finished && tid == 0) atomicAdd(&W.counters[kTeamLeft], 1);
    }
    if (tid == 0) {
        if (n_f) atomicAdd(&W.counters[kTotals + 0], n_f);
        if (n_g) atomicAdd(&W.counters[kTotals + 1], n_g);
        if (n_q) atomicAdd(&W.counters[kTotals + 2], n_q);
    }
#endif
}

// before: the queue head and the give-up count at zero; after: the counters as a finished lock-step solve leaves them
// (slot 0 empty, the seed list's -count undone), the give-up count where mpc_solve_result reports "unfinished"
__global__ void team_open_kernel(SolveWorkspace W)
{
    W.counters[4] = 0;
    W.counters[kTeamLeft] = 0;
}
__global__ void team_close_kernel(SolveWorkspace W, int count)
{
    W.counters[0] = W.counters[kTeamLeft];
    W.counters[1] = 0;
    W.counters[2] = 0;
    W.counters[3] = W.counters[kTeamLeft];
    W.counters[kTotals] += count;
}

}  // namespace

namespace mpck {

namespace {
// dynamic LDS of one team: the QP's pool (augmented layout, the larger) or, between QPs, the body's vectors and one
// scenario of the evaluation behind them, whichever is more; 0: not supported
size_t team_lds_bytes(const mpc_params &P)
{
    if (!prefix_nz_supported(P)) return 0;
    const int H = P.horizon, n = 6 * H, m = 12 * (H + 1);
    const size_t qp = lds_bytes(H, 1);
    if (qp == 0 || 6 * ((H + 1) / 2) > 64) return 0;               // the evaluation: one wavefront per scenario
    const size_t eval = sizeof(double) * ((size_t)((6 * n + 3 * m + 1) & ~1) + (size_t)((H + 1) & ~1) +
                                          (size_t)make_layout(H, kCompact, P.dt_cost == P.dt_con).total);
    const size_t need = eval > qp ? eval : qp;
    return need <= kQpLdsLimit / 2 ? need : 0;                       // two teams per CU
}
}  // namespace

bool solve_team_supported(const mpc_params &P) { return team_lds_bytes(P) != 0; }

// Runs every problem of a solve that mpc_solve_begin has set up (and no round has touched) to its end.  Afterwards
// launch_solve_result(..., slot 0) reports as after lock-step rounds; its "unfinished" counts the problems that met the
// bound on body calls (none in practice: 64 per allowed iteration).
hipError_t launch_solve_team(const mpc_params &P, const double *decay, const mpc_scenarios &sc, const SolveWorkspace &W, int count,
                             int maxiter, int teams, double *work, int *iwork, hipStream_t stream)
{
    if (!solve_team_supported(P)) return hipErrorNotSupported;
    if (count <= 0) return hipSuccess;
    const Geometry g = geometry(P, kCompact);
    PrefixArgs A;
    A.decay = decay; A.sc = sc; A.x = nullptr; A.cost = nullptr; A.grad = nullptr; A.con = nullptr; A.jac = nullptr;
    A.compact = nullptr; A.cstride = (int)jacobian_nz_count(P.horizon); A.tmajor = 0;
    A.S = 1; A.items = g.items;
    A.list = nullptr; A.count_dev = nullptr;
    const size_t lds = team_lds_bytes(P);
    const int t = teams < count ? teams : count;
    const int max_calls = 64 * (maxiter + 2);
    hipLaunchKernelGGL(team_open_kernel, dim3(1), dim3(1), 0, stream, W);
    const bool same_dt = P.dt_cost == P.dt_con;
    auto go = [&](auto kernel) -> hipError_t {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kernel, dim3((unsigned)t), dim3(kQpThreads), lds, stream, P, A, W, count, max_calls, work, iwork,
                           qp_work_doubles(P.horizon), qp_work_ints(P.horizon));
        return hipGetLastError();
    };
    hipError_t e = same_dt ? go(solve_team_kernel<true>) : go(solve_team_kernel<false>);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(team_close_kernel, dim3(1), dim3(1), 0, stream, W, count);
    return hipGetLastError();
}

}  // namespace mpck
