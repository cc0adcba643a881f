/*
 * mpc_amd.h — C-ABI of libmpc_amd.so: the UR20 MPC inner numeric loop on MI355X (gfx950).
 *
 * The reference (canongun/MPC) has no FFI; its "operator API" for this path is a set of
 * Python callables handed to scipy.optimize.minimize (ur_mpc.py:188-200).  Each entry point
 * below names the reference callable it replaces.  Paths are relative to
 * /root/reference/ur20_mpc_controller/ur20_mpc_controller/models/ :
 *   MPC:<line> = ur_mpc.py, DYN:<line> = ur20_dynamics.py; NUMDIFF = scipy/optimize/_numdiff.py,
 *   SLSQP = scipy/optimize/_slsqp_py.py (scipy 1.15.3, the third-party caller of the callables).
 *
 * Conventions
 *   - all array arguments are DEVICE pointers (e.g. torch tensor.data_ptr()), fp64, owned by the
 *     caller, laid out [N][...] with the scenario index slowest (each scenario's row contiguous);
 *   - `stream` is a hipStream_t passed as void*; work is enqueued asynchronously on it;
 *   - every function returns MPC_OK (0) or a negative mpc_status; none throws;
 *     mpc_last_error() gives the text of the last failure on that context;
 *   - one context per caller thread and device; a context is not re-entrant.
 *
 * Sizes: H = horizon, n = 6*H decision variables, m = 12*(H+1) inequality rows.
 */
#ifndef MPC_AMD_H
#define MPC_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MPC_AMD_ABI_VERSION 11
#define MPC_MAX_HORIZON 255

typedef enum mpc_status {
    MPC_OK = 0,
    MPC_ERR_INVALID_ARG = -1,
    MPC_ERR_HIP = -2,
    MPC_ERR_UNSUPPORTED = -3,
    MPC_ERR_NO_PARAMS = -4,
    MPC_ERR_NO_DEVICE = -5
} mpc_status;

typedef enum mpc_algo {
    MPC_ALGO_AUTO = 0,
    /* one workgroup per (scenario, perturbation group), one lane per horizon step, full rollouts,
     * strict (uncontracted) fp64 arithmetic in the oracle's evaluation order */
    MPC_ALGO_ROLLOUT = 1,
    /* one workgroup per scenario: base rollout once, then every perturbed rollout restarts from
     * the shared prefix at the perturbed step */
    MPC_ALGO_PREFIX = 2
} mpc_algo;

typedef struct mpc_ctx mpc_ctx;

/* Everything URMPC.__init__ / UR20Dynamics.__init__ read from the parameter server
 * (MPC:29-81, DYN:18-47) plus the kinematic table that stands in for MoveIt (DYN:82). */
typedef struct mpc_params {
    int32_t horizon;        /* MPC:29 */
    int32_t algo;           /* mpc_algo */
    double dt_cost;         /* UR20Dynamics.dt, step of the rollout inside _cost_function (MPC:686, DYN:19,97-98) */
    double dt_con;          /* URMPC.dt, step of predict_trajectory inside the constraint (MPC:30,705) */
    double w_position;      /* MPC:39 */
    double w_orientation;   /* MPC:40 */
    double w_control;       /* MPC:41 */
    double w_smooth;        /* MPC:42 */
    double w_damping;       /* MPC:59 */
    double w_linear;        /* MPC:45 */
    double w_angular;       /* MPC:46 */
    double terminal_factor; /* 2.0, MPC:373 */
    double fd_eps;          /* 1e-3, options['eps'] MPC:197 -> abs_step of NUMDIFF:498-524 */
    double q_lo[6];         /* DYN:31-38 */
    double q_hi[6];
    double v_lim[6];        /* DYN:40-47; also the bounds on x, MPC:420-430 */
    /* standard-DH table of the arm; ca/sa = cos/sin of the link twists (exact 0/+-1 for a UR) */
    double dh_a[6], dh_d[6], dh_ca[6], dh_sa[6];
    double base_c, base_s;  /* yaw of the DH base in the frame the Jacobian is expressed in */
    double tool[3];         /* tool0 -> tip translation, in the tool0 frame */
} mpc_params;

/* Per-scenario inputs of one batch: what compute_control assembles before it calls minimize
 * (MPC:148-157).  Rows: state0 = [q, dq, ee_pos, ee_rpy] (DYN:49-56); target = [pos, rpy];
 * base_traj[k] = [position, orientation, linear_velocity, angular_velocity] of
 * predict_base_motion's entry k (MPC:592-597,645-650); Jc = _J_cached row-major (MPC:157);
 * u_prev/has_prev = previous_control and whether the attribute exists yet (MPC:209,313). */
typedef struct mpc_scenarios {
    int32_t count;             /* N */
    int32_t reserved;
    const double *state0;      /* [N][18] */
    const double *target;      /* [N][6]  */
    const double *base_traj;   /* [N][H][12] */
    const double *Jc;          /* [N][36] */
    const double *u_prev;      /* [N][6] or NULL */
    const uint8_t *has_prev;   /* [N] or NULL (= all zero) */
} mpc_scenarios;

/* version of this ABI (MPC_AMD_ABI_VERSION of the library that was loaded) */
int mpc_abi_version(void);

/* Create a context bound to HIP device `device` for horizon `horizon` and up to `max_scenarios`
 * scenarios per call (sizes the internal workspace).  Replaces URMPC.__init__ (MPC:15-81). */
int mpc_ctx_create(int device, int horizon, int max_scenarios, mpc_ctx **out);
void mpc_ctx_destroy(mpc_ctx *ctx);

/* Weights, limits, kinematic table and the decay table decay[k] = 0.95**k, k < H (MPC:294),
 * evaluated by the host so that it is bit-identical to CPython's pow. */
int mpc_set_params(mpc_ctx *ctx, const mpc_params *params, const double *decay_host, int n_decay);

/* f(x) and c(x) for N scenarios: replaces URMPC._cost_function (MPC:665-692) and the
 * dynamics_constraint closure (MPC:388-411), i.e. SLSQP's mode-1 request (SLSQP:434-436).
 *   x [N][n] -> cost [N], con [N][m] (con may be NULL). */
int mpc_eval(mpc_ctx *ctx, const mpc_scenarios *sc, const double *x, double *cost, double *con,
             void *stream);

/* One "eval" per scenario: f, c, the forward-difference gradient and constraint Jacobian that
 * scipy builds around the two callables (approx_derivative '2-point', abs_step = fd_eps, step
 * sign flipped at the upper bound: NUMDIFF:44-72,498-524,584-625; wiring SLSQP:299-313,381-387),
 * i.e. SLSQP's mode -1 request (SLSQP:438-440).
 *   x [N][n] -> cost [N], grad [N][n], con [N][m],
 *   con_jac [N][n][m]: per scenario COLUMN-major m x n (con_jac[s][i][r] = d c_r / d x_i), the
 *   layout the Fortran core of SLSQP consumes (a(la, n+1), SLSQP:509).  grad/con/con_jac may be NULL. */
int mpc_eval_fd(mpc_ctx *ctx, const mpc_scenarios *sc, const double *x, double *cost, double *grad,
                double *con, double *con_jac, void *stream);

/* mpc_eval_fd for callers that cannot allocate (code under HIP graph capture, tight serving loops, harnesses
 * that forbid hipMalloc).  NO COUNTERPART IN THE REFERENCE: same arguments, same outputs, always the
 * MPC_ALGO_ROLLOUT kernels (bit-identical to mpc_eval_fd on a context whose algo is MPC_ALGO_ROLLOUT; any 6R
 * DH table), but the per-rollout costs between its two kernels live in a file-scope device array of
 * MPC_STATIC_SCRATCH_DOUBLES doubles (16 MiB, held by every GPU the library runs on until the process ends)
 * instead of a context allocation.  The call itself allocates and frees nothing.  Limit: count * (6H + 1) <=
 * MPC_STATIC_SCRATCH_DOUBLES (17,331 scenarios at H = 20, 3,489 at H = 100), MPC_ERR_INVALID_ARG beyond it.
 * All calls on one GPU share that GPU's array: calls that overlap on one GPU (two streams, two contexts, two
 * threads) race; nothing in it has to start at zero (each launch writes every entry it reads). */
#define MPC_STATIC_SCRATCH_DOUBLES (2 * 1024 * 1024)
int mpc_eval_fd_static(mpc_ctx *ctx, const mpc_scenarios *sc, const double *x, double *cost, double *grad,
                       double *con, double *con_jac, void *stream);

/* Structured form of the same forward-difference dc/dx.  Column x_{6j+a} (control of joint a at step j)
 * only reaches the rows of joint a: the velocity row of state j+1 and the position rows of states
 * j+1..H (MPC:398-406 over the Euler rollout DYN:101-128); every other entry of the dense matrix is
 * an exact zero.  con_jac_nz [N][mpc_jacobian_nz_count(H)] holds those quotients:
 *   nz[6j + a]                  = d c[12(j+1) + 6 + a] / d x[6j + a]            (j = 0..H-1)
 *   nz[6H + 6 T(j,k) + a]       = d c[12k + a]         / d x[6j + a]            (k = j+1..H),
 *   T(j,k) = jH - j(j-1)/2 - j - 1 + k   (pairs (j,k) in lexicographic order).
 * Same numbers as mpc_eval_fd's con_jac, 22x fewer bytes at H=20 (11 KB instead of 242 KB per scenario):
 * the form a caller that expands on its own side (the batched SLSQP driver) takes across PCIe.
 * Needs the UR twist pattern in the DH table (MPC_ERR_UNSUPPORTED otherwise).  grad/con may be NULL. */
int mpc_jacobian_nz_count(int horizon);
int mpc_eval_fd_nz(mpc_ctx *ctx, const mpc_scenarios *sc, const double *x, double *cost, double *grad,
                   double *con, double *con_jac_nz, void *stream);

/* con_jac_nz [N][nz] -> dense con_jac [N][n][m] (the layout of mpc_eval_fd), on the device. */
int mpc_expand_jacobian(mpc_ctx *ctx, int count, const double *con_jac_nz, double *con_jac, void *stream);

/* Batched 6x6 geometric Jacobian, row-major, rows 0-2 linear / 3-5 angular: replaces
 * UR20Dynamics.get_jacobian (DYN:74-87; MoveIt's getJacobian in the reference).  q [N][6] -> J [N][36]. */
int mpc_jacobian(mpc_ctx *ctx, int count, const double *q, double *J, void *stream);

/* Batched forward kinematics of the same chain: tip position and roll/pitch/yaw ('sxyz'), what
 * the action server reads from MoveIt to fill ee_pose (mpc_action_server.py:57-75).
 * q [N][6] -> pose [N][6] = [x,y,z,roll,pitch,yaw]. */
int mpc_fk(mpc_ctx *ctx, int count, const double *q, double *pose, void *stream);

/* Test hook, no counterpart in the reference (which calls numpy's sin/cos, DYN:97-128 through the DH chain):
 * the evaluation kernels' own sin/cos routine applied to x [count] -> sn, cs [count], so that its accuracy over
 * the joint range is tested directly.  Needs no parameters set. */
int mpc_sincos_probe(mpc_ctx *ctx, int count, const double *x, double *sn, double *cs, void *stream);

/* Test hook, no counterpart in the reference: the ordered sum that the solver kernels run as one chain of FP64
 * additions on a wavefront (mpc_amd/csrc/lsq_wave.hpp, ordered_sum_from; a computed jump into a block of 128 DPP
 * instructions).  Case i: out[i] = ((t0[i] + terms[i][first[i]]) + terms[i][first[i] + 1]) + ... + terms[i][127],
 * terms [count][128], first[i] in 0..128 (128: nothing is added).  Needs no parameters set. */
int mpc_ordered_sum_probe(mpc_ctx *ctx, int count, const double *terms, const double *t0, const int32_t *first,
                          double *out, void *stream);

/* Batched platform predictor: replaces URMPC._predict_base_dynamic (MPC:606-651).
 * base [N][12] = [position, orientation, linear_velocity, angular_velocity] (base_observer.py:55-62),
 * dyn_host[6] = {mass, I_z, k_v, k_omega, tau_linear, tau_angular} (HOST pointer) -> base_traj [N][H][12]. */
int mpc_predict_base_dynamic(mpc_ctx *ctx, int count, const double *base, const double *dyn_host,
                             double *base_traj, void *stream);
/* The same with the commanded velocities of the actuator-lag model, the optional 'v_cmd' / 'omega_cmd' keys of the
 * platform state (MPC:624-625): cmd [N][2] = [v_cmd, omega_cmd] per row, or NULL (then every row's commands are its
 * current v and omega, i.e. mpc_predict_base_dynamic). */
int mpc_predict_base_dynamic_cmd(mpc_ctx *ctx, int count, const double *base, const double *cmd, const double *dyn_host,
                                 double *base_traj, void *stream);

/* Everything URMPC.compute_control assembles before it calls minimize (MPC:104-185), for N controllers
 * in one launch, device to device:
 *   active[s]    = 0 where both platform velocities are below 0.01 (the early exit, MPC:108), else 1
 *   state0[s]    = [q, dq (zeros when dq is NULL, as mpc_action_server.py:83), ee_pose]   (MPC:148-153)
 *   base_traj[s] = _predict_base_dynamic(base[s])                                         (MPC:606-651)
 *   Jc[s]        = J(q[s])                                                                (MPC:157)
 *   u_init[s]    = J^T (J J^T + 0.01 I)^-1 twist, the damped least-squares start          (MPC:112-143,168-175)
 *   x0[s]        = previous solution shifted by one step with u_init appended where has_solution[s],
 *                  else u_init repeated H times (MPC:177-185); x0 / prev_solution / has_solution may be NULL;
 *                  x0 must not overlap prev_solution (no in-place shift: MPC_ERR_INVALID_ARG).
 * q [N][6], ee_pose [N][6] = [pos, rpy], base [N][12], dyn_host as in mpc_predict_base_dynamic.
 * The 6x6 solve is Gaussian elimination with partial pivoting in fp64; it agrees with LAPACK's to
 * rounding, not bit for bit. */
int mpc_prepare_tick(mpc_ctx *ctx, int count, const double *q, const double *dq, const double *ee_pose,
                     const double *base, const double *dyn_host, const double *prev_solution,
                     const uint8_t *has_solution, double *state0, double *base_traj, double *Jc, double *u_init,
                     double *x0, uint8_t *active, void *stream);

/* The QP subproblem of one SLSQP major iteration (what scipy's compiled core spends its time in; called
 * from _slsqp_py.py:429 via the Fortran LSQ) for `count` problems, one workgroup each:
 *   minimise || e s - f ||, e = sqrt(d) l^T, f = -d^(-1/2) l^(-1) g   (l: packed LDL' factor of the BFGS matrix,
 *   [(n+1)n/2 + 1] per problem, g [n] the gradient), subject to c + (dc/dx) s >= 0 (c [m], dc/dx in the
 *   structured form nz of mpc_eval_fd_nz) and lo <= s <= hi ([n] each: the box around the iterate).
 * Outputs: s [count][n] the step, y [count][m + 2n] the multipliers (constraint rows first), mode [count]
 * (1 = solved, otherwise the code the SLSQP body hands on, 4 = inconsistent linearisation).  The arithmetic
 * is the host core's (mpc_amd/csrc/slsqp_core.c), entry for entry, so results are bit-identical to it and to
 * scipy's.  All pointers are device pointers.  Needs a context whose horizon matches.  The first call on a context
 * allocates the workspace of its resident workgroups (two per CU while the packed triangle and the vectors fit
 * LDS, H <= 26: 0.95 MB each at H = 20, 0.5 GB per context on MI355X); like every call on one context, QP calls
 * must not overlap each other (one workspace): concurrent callers use one context each. */
int mpc_qp_solve(mpc_ctx *ctx, int count, const double *l, const double *g, const double *c, const double *nz,
                 const double *lo, const double *hi, double *s, double *y, int32_t *mode, void *stream);

/* One major-iteration step of SLSQP solves whose QP side lives on the device (include/mpc_slsqp.h, QP offload),
 * for the `count` problems listed in index (int32, device, each in [0, rows)): per problem, in this order,
 *   flags bit 0: reset the LDL' factor of the BFGS matrix to the identity        (the body's "reset BFGS matrix"),
 *   flags bit 1: BFGS-update it with the new derivatives g, nz, the last step s_in, the multipliers and the
 *                Lagrangian gradient kept from the previous call                  (the body's block after mode -1),
 *   solve the QP (as mpc_qp_solve) with the box lo, hi, and keep r and v = g - (dc/dx)^T r for the next update.
 * Per-problem state, `rows` rows each, owned by the caller and only touched by this call: L [(n+1)n/2 + 1],
 * R [m + 2n], V [n]; inputs per problem G [n], C [m], NZ [6H + 3H(H+1)] (what mpc_eval / mpc_eval_fd_nz returned at
 * the problem's current iterate).  Per request (position k in the list): flags, lo, hi, s_in [n] in;
 * s_out [n], r_out [m] (constraint multipliers), mode out.  Bit-identical to the host core.
 * The indices must be distinct (two requests for one problem would race on its state).  A request whose index lies
 * outside [0, rows) touches no state and gets mode -100.  A request with flags bit 2 (Kraft's augmented problem for an
 * inconsistent linearisation, mpc_amd/csrc/slsqp_core.h) is not served by this call and gets mode -101: the
 * device-resident solve below serves such requests, a host-side caller solves that problem with the host driver. */
int mpc_qp_step(mpc_ctx *ctx, int count, int rows, const int32_t *index, const int32_t *flags, const double *lo,
                const double *hi, const double *s_in, double *L, const double *G, const double *C, const double *NZ, double *R,
                double *V, double *s_out, double *r_out, int32_t *mode, void *stream);

/* Whole SLSQP solves resident on the device: replaces scipy's driver loop around its compiled core
 * (SLSQP:427-456, called from MPC:188-200) for sc->count problems at once, with nothing but an "unfinished" count
 * crossing PCIe between the start point going in and the solutions coming out.  Per round and problem: one call of
 * Kraft's SLSQP body (merit function, line search, convergence tests: the text of mpc_amd/csrc/slsqp_body.inc, the same
 * file the host core of libmpc_slsqp.so is compiled from) as a one-wavefront-per-problem kernel, which sorts the
 * unfinished problems into three device-side request lists; then f and c (mpc_eval), grad f and the structured dc/dx
 * (mpc_eval_fd_nz) and the BFGS update + QP (mpc_qp_step) for the problems on each list, all reading the lists and
 * their lengths from device memory.  Iterates, exit modes and iteration counts are those of the host drivers, bit for bit.
 *   ws: caller-owned device workspace of mpc_solve_workspace_bytes(H, rows) bytes, 256-byte aligned, rows >= sc->count
 *       (85 KB per row at H = 20); it holds every problem's state between the calls.
 *   mpc_solve_begin:  x0 [N][n] is clipped into the box +-v_lim (SLSQP:268, MPC:420-430), the state of a first call
 *       is set up (acc = ftol, maxiter: MPC:195-196) and f, c, grad f, dc/dx are evaluated at the start point
 *       (SLSQP:419-425).  Also sizes the context's QP workspace, so that mpc_solve_rounds never allocates.
 *   mpc_solve_rounds: enqueues `rounds` rounds.  `bound` >= the number of unfinished problems sizes the grids (pass
 *       sc->count, or the count a previous call reported).  After the last round the number of problems that were
 *       still unfinished in it is copied to *unfinished (device or pinned host memory, may be NULL) on the stream;
 *       0 means every problem has stopped; with bound == 0 nothing is launched but that copy.  -1 means a round met
 *       more requests than `bound` sized its grids for (problems were left out: the solve is void, begin again).
 *       If a launch fails part-way the call returns MPC_ERR_HIP and the workspace is forgotten: further rounds and
 *       mpc_solve_result are refused until mpc_solve_begin has run again.  A problem whose linearisation
 *       is inconsistent (QP mode 4) goes on with Kraft's augmented problem on the device, as in the host core
 *       (slsqp_body.inc: one more variable, weight 100 .. 1e7): such QPs are solved by a kernel of their own, which
 *       the rounds launch from the first call on that finds the flag up which the QP kernel raises (a word of
 *       pinned host memory) when it meets one; until then the problem waits, asking again every round.
 *   mpc_solve_result: x [N][n], fun [N], nit [N], status [N] (SLSQP's exit mode, 0 = converged) and, if non-NULL,
 *       stats[5] = {mpc_eval requests, mpc_eval_fd_nz requests, QP steps, unfinished in the last round, QP steps that
 *       ended with the unconstrained step (no constraint near: the QP kernel skips the transformed matrix and NNLS)}
 *       (device).
 * `sc` must be the same batch in all three calls; calls on one context / one workspace must not overlap. */
size_t mpc_solve_workspace_bytes(int horizon, int rows);
int mpc_solve_begin(mpc_ctx *ctx, const mpc_scenarios *sc, const double *x0, double acc, int maxiter, void *ws, int rows,
                    size_t ws_bytes, void *stream);
int mpc_solve_rounds(mpc_ctx *ctx, const mpc_scenarios *sc, void *ws, int rows, size_t ws_bytes, int rounds, int bound,
                     int32_t *unfinished, void *stream);
int mpc_solve_result(mpc_ctx *ctx, int count, void *ws, int rows, size_t ws_bytes, double *x, double *fun, int32_t *nit,
                     int32_t *status, int32_t *stats, void *stream);
/* The same solves without the lock step, for batches too small to fill the rounds (a round costs at least one QP's
 * latency however few problems ask for one; a single controller's tick is ~100 rounds of five launches): ONE launch in
 * which persistent workgroups take a problem each from a queue and run it to its end - body, evaluations, BFGS update +
 * QP, Kraft's augmented problem included - with the device code of the lock-step kernels on the same workspace rows,
 * so results are those of mpc_solve_rounds bit for bit.  Call it after mpc_solve_begin instead of mpc_solve_rounds
 * (same sc, ws; maxiter as given to mpc_solve_begin: it bounds the work per problem), then mpc_solve_result.
 * *unfinished (device or pinned host memory, may be NULL) receives the number of problems that met that bound (0).
 * H <= 21 (the QP's vectors in LDS, one wavefront per evaluation); MPC_ERR_UNSUPPORTED otherwise.  At H = 20 one
 * controller's solve takes ~5 ms instead of ~9, 4,096 take ~65 ms instead of 106; from ~16,384 on the rounds win. */
int mpc_solve_run(mpc_ctx *ctx, const mpc_scenarios *sc, void *ws, int rows, size_t ws_bytes, int maxiter, int32_t *unfinished,
                  void *stream);

/* Active-set gather for lock-step solves: scipy's driver evaluates a problem only when its SLSQP core
 * asks (mode 1 / -1, _slsqp_py.py:427-456), so each round of a batched solve touches a changing subset
 * of the scenarios.  Copies rows index[i] (device int32[count], each in [0, src->count)) of every array
 * of `src` to row i of `dst` in one launch; optional arrays are copied when both sides have them.
 * Rows whose index is out of range are filled with NaN (never read out of bounds).  dst->count is
 * ignored (dst must hold `count` rows). */
int mpc_gather_scenarios(mpc_ctx *ctx, const mpc_scenarios *src, const int32_t *index, int count,
                         const mpc_scenarios *dst, void *stream);

/* Text of the last error recorded on ctx (or on the calling thread when ctx is NULL). */
const char *mpc_last_error(const mpc_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* MPC_AMD_H */
