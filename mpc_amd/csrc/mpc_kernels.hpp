// Host-side launchers of the gfx950 kernels (one translation unit per arithmetic mode).
#pragma once

#include <hip/hip_runtime.h>

#include "mpc_amd.h"

struct slsqp_state;

namespace mpck {

// MPC_ALGO_ROLLOUT: one workgroup per (scenario, perturbation group), one lane per horizon step.
// Writes f of rollout i (0 = unperturbed, v+1 = variable v stepped) to fvals[s*fstride + i];
// the unperturbed rollout writes con[s][m]; perturbed rollout v writes column v of jac[s][n][m].
hipError_t launch_rollout(const mpc_params &P, const double *decay, const mpc_scenarios &sc, const double *x,
                          double *fvals, int fstride, double *con, double *jac, int n_pert,
                          hipStream_t stream);

// grad[s][v] = (fvals[s][v+1] - fvals[s][0]) / dx_v, cost[s] = fvals[s][0]  (NUMDIFF:594-619)
hipError_t launch_fd_finalize(const mpc_params &P, int N, const double *x, const double *fvals, double *cost,
                              double *grad, hipStream_t stream);

// MPC_ALGO_PREFIX: one workgroup per scenario, shared-prefix perturbed rollouts.
// `workspace` (device, `workspace_doubles` long) is only used for long horizons; its size per
// scenario is prefix_workspace_doubles_per_scenario() (0 when the fused kernel serves P).
// `pipe` (optional) lets the long-horizon path run the expansion of one chunk of scenarios on a
// second stream while the next chunk's rollouts run on `stream`; `stream` waits for the last
// expansion before launch_prefix returns, so callers see ordinary stream ordering.
struct PrefixPipeline {
    hipStream_t aux = nullptr;
    hipEvent_t filled[2] = {nullptr, nullptr};    // half h of the workspace holds a chunk's quotients
    hipEvent_t drained[2] = {nullptr, nullptr};   // the expansion has finished reading half h
};
hipError_t launch_prefix(const mpc_params &P, const double *decay, const mpc_scenarios &sc, const double *x,
                         double *cost, double *grad, double *con, double *jac, double *workspace,
                         size_t workspace_doubles, hipStream_t stream, const PrefixPipeline *pipe = nullptr);
// Structured dc/dx (include/mpc_amd.h, mpc_eval_fd_nz): the rollout kernel alone, quotients to `nz`.
hipError_t launch_prefix_nz(const mpc_params &P, const double *decay, const mpc_scenarios &sc, const double *x,
                            double *cost, double *grad, double *con, double *nz, hipStream_t stream);
hipError_t launch_expand(const mpc_params &P, int count, const double *nz, double *jac, hipStream_t stream);
bool prefix_nz_supported(const mpc_params &P);
size_t jacobian_nz_count(int horizon);
bool prefix_supported(const mpc_params &P);
size_t prefix_workspace_doubles_per_scenario(const mpc_params &P);

hipError_t launch_jacobian(const mpc_params &P, int N, const double *q, double *J, hipStream_t stream);
hipError_t launch_fk(const mpc_params &P, int N, const double *q, double *pose, hipStream_t stream);
hipError_t launch_predict_base_dynamic(int N, int H, double dt, const double *base, const double *cmd, const double (&dyn)[6],
                                       double *out, hipStream_t stream);

// MPC:104-185 for N controllers: early-exit flag, state0, platform prediction, Jc, DLS start, start point
hipError_t launch_prepare_tick(const mpc_params &P, int N, const double (&dyn)[6], const double *q, const double *dq,
                               const double *ee_pose, const double *base, const double *prev_solution,
                               const uint8_t *has_solution, double *state0, double *base_traj, double *Jc,
                               double *u_init, double *x0, uint8_t *active, hipStream_t stream);

// SLSQP's QP subproblem, one workgroup per problem (kernels_qp.hip).  `teams` workgroups share the batch;
// work / iwork hold teams * qp_work_doubles(H) doubles / teams * qp_work_ints(H) ints.
#ifndef MPC_QP_THREADS
#define MPC_QP_THREADS 256
#endif
constexpr int kQpThreads = MPC_QP_THREADS;
constexpr size_t kQpLdsLimit = 160 * 1024;   // LDS of one CU (gfx950)
int qp_teams_per_cu(int H);                  // resident teams per CU the QP kernels are sized for
size_t qp_work_doubles(int H);
size_t qp_work_ints(int H);
hipError_t launch_qp(int H, int count, int teams, const double *l, const double *g, const double *c, const double *nz,
                     const double *lo, const double *hi, double *s, double *y, int32_t *mode, double *work, int *iwork,
                     hipStream_t stream);

#define MPC_QP_MODE_BAD_INDEX (-100)   // mpc_qp_step: the request's problem index lies outside [0, rows)
#define MPC_QP_MODE_DEFERRED (-101)    // an augmented request (MPC_QP_FLAG_AUG) met the plain instance: not served yet
hipError_t launch_qp_step(int H, int count, int teams, int rows, const int32_t *index, const int32_t *flags, const double *lo,
                          const double *hi, const double *s_in, double *L, const double *G, const double *C, const double *NZ,
                          double *R, double *V, double *s_out, double *r_out, int32_t *mode, double *work, int *iwork,
                          hipStream_t stream);

// ---- whole SLSQP solves resident on the device (kernels_slsqp.hip; mpc_solve_* in include/mpc_amd.h) ----------
// The caller's workspace carved into per-problem rows (`rows` of them): iterate, evaluation point, line-search
// origin, step, box, penalty multipliers, QP multipliers, f, c, grad f, structured dc/dx, LDL' factor, Lagrangian
// gradient, the body's saved scalars, request lists and counters.
struct SolveWorkspace {
    int rows;
    size_t bytes;
    double *lb, *ub;
    double *X, *XE, *X0, *S, *LO, *HI, *MU, *RM, *F, *C, *G, *NZ, *L, *V, *ACC;
    ::slsqp_state *ST;
    int32_t *MODE, *ITER, *QPMODE, *QPFLAGS, *LIST_F[2], *LIST_G[2], *LIST_Q[2], *counters;   // lists, counters: two slots
};
SolveWorkspace carve_solve_workspace(void *base, int H, int rows);
size_t solve_workspace_bytes(int H, int rows);
hipError_t launch_solve_begin(const mpc_params &P, const SolveWorkspace &W, int count, const double *x0, double acc, int maxiter,
                              hipStream_t stream);
// round r = 0, 1, ...: reads list buffer r % 2 and counter slot r % 3, fills buffer (r + 1) % 2 and slot (r + 1) % 3
hipError_t launch_solve_body(int H, const SolveWorkspace &W, int bound, int round, hipStream_t stream);
hipError_t launch_solve_result(int H, const SolveWorkspace &W, int count, int slot, double *x, double *fun, int32_t *nit,
                               int32_t *status, int32_t *stats, hipStream_t stream);
hipError_t launch_solve_count(const SolveWorkspace &W, int slot, hipStream_t stream);
// kernels_team.hip: every problem run to its end by one workgroup at a time (no lock step), after launch_solve_begin
bool solve_team_supported(const mpc_params &P);
hipError_t launch_solve_team(const mpc_params &P, const double *decay, const mpc_scenarios &sc, const SolveWorkspace &W, int count,
                             int maxiter, int teams, double *work, int *iwork, hipStream_t stream);
hipError_t launch_ordered_sum_probe(int count, const double *terms, const double *t0, const int32_t *first, double *out,
                                    hipStream_t stream);
// One request list of a device-resident solve: evaluate rows list[0 .. *count) of the per-problem arrays in place
// (x from XE, outputs to the problem's own rows).  `bound` >= *count sizes the grid; surplus workgroups exit.
struct RowList {
    const int32_t *list = nullptr;    // device, row numbers
    const int32_t *count = nullptr;   // device, how many
    int bound = 0;
    int32_t *head = nullptr;          // device, zero on entry: work-queue head of the QP kernel (teams take requests in turn)
};
// augmented = false: the plain instance (augmented requests are answered MPC_QP_MODE_DEFERRED and *aug_seen, which may
// be host memory the device can write, is set to 1); true: the instance that serves only the augmented requests.
hipError_t launch_sincos_probe(int n, const double *x, double *sn, double *cs, hipStream_t stream);
hipError_t launch_prefix_rows(const mpc_params &P, const double *decay, const mpc_scenarios &sc, const double *x, double *cost,
                              double *grad, double *con, double *nz, const RowList &rows, hipStream_t stream);
hipError_t launch_qp_step_rows(int H, int teams, int rows, const RowList &list, const int32_t *flags, const double *lo,
                               const double *hi, double *S, double *L, const double *G, const double *C, const double *NZ,
                               double *R, double *V, int32_t *mode, double *work, int *iwork, int32_t *aug_seen, bool augmented,
                               int32_t *free_steps, hipStream_t stream);
constexpr int kSolveFreeSteps = 29;    // place in a solve workspace's counters: QPs that ended with lsq's free step

// dst row i = src row index[i] for every scenario array (NaN rows for an index outside [0, src.count))
hipError_t launch_gather_scenarios(int H, const mpc_scenarios &src, const int32_t *index, int count,
                                   const mpc_scenarios &dst, hipStream_t stream);

// Blocks b and b+8 share an XCD (round-robin dispatch): give every XCD one contiguous range of
// logical ids so that the workgroups of one scenario hit the same L2.  Bijective for any grid.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nwg)
{
    const unsigned xcd = bid & 7u, q = nwg >> 3, r = nwg & 7u;
    const unsigned start = (xcd < r) ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q;
    return start + (bid >> 3);
}

}  // namespace mpck
