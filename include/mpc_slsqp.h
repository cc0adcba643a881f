/* libmpc_slsqp.so: native host driver for MANY SLSQP solves advancing together (no Python, no GIL).
 *
 * The reference solves one problem per tick with scipy.optimize.minimize(method='SLSQP')
 * (ur_mpc.py:188-200).  scipy's driver (_slsqp_py.py:216-456) is a reverse-communication loop around a
 * compiled core that asks for "f and c" (mode 1) or "grad f and dc/dx" (mode -1) until it is done.  This
 * library keeps the state of `capacity` such problems in C, advances every unfinished one by one core
 * call per round on `threads` host threads (mpc_slsqp_advance), and tells the caller which problems
 * asked for what; the caller fills those rows (from libmpc_amd's mpc_eval / mpc_eval_fd_nz) and calls
 * again.  The core is a C restatement of the published algorithm (mpc_amd/csrc/slsqp_core.c) pinned bit
 * for bit against scipy's (tests/test_native_slsqp.py), so iterates equal the reference's.
 *
 * Problem shape is the MPC's: n = 6H variables, m = 12(H+1) inequality rows, box bounds, no equalities.
 * dc/dx is handed over in its structured form (include/mpc_amd.h, mpc_eval_fd_nz) and expanded into the
 * core's dense a(m, n+1) on the worker thread.  Host pointers only; nothing here touches the GPU. */
#ifndef MPC_SLSQP_H
#define MPC_SLSQP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mpc_slsqp_batch mpc_slsqp_batch;

#define MPC_SLSQP_ABI_VERSION 2
int mpc_slsqp_abi_version(void);

/* threads <= 0: as many as OpenMP would use (OMP_NUM_THREADS).  Returns 0 or a negative code. */
int mpc_slsqp_create(int horizon, int capacity, int threads, mpc_slsqp_batch **out);
void mpc_slsqp_destroy(mpc_slsqp_batch *b);

/* Start `count` problems: x0 [count][n] is clipped to [lb, ub] (_slsqp_py.py:268), acc = ftol,
 * maxiter as scipy's option (_slsqp_py.py:236-241,404).  Every problem then wants f, c, grad f, dc/dx
 * at its start point (_slsqp_py.py:419-425): fill ALL rows before the first mpc_slsqp_advance. */
int mpc_slsqp_begin(mpc_slsqp_batch *b, int count, const double *x0, const double *lb, const double *ub, double acc,
                    int maxiter);

/* Rows the caller reads (x) and fills (f, c, g, nz); all [capacity][...] and owned by the batch:
 * x [.][n] current iterate (evaluate at clip(x, lb, ub), _slsqp_py.py:435), f [.], c [.][m],
 * g [.][n+1] grad f in the first n entries of each row (the last belongs to the core),
 * nz [.][6H + 3H(H+1)] structured dc/dx. */
double *mpc_slsqp_x(mpc_slsqp_batch *b);
double *mpc_slsqp_f(mpc_slsqp_batch *b);
double *mpc_slsqp_c(mpc_slsqp_batch *b);
double *mpc_slsqp_g(mpc_slsqp_batch *b);
double *mpc_slsqp_nz(mpc_slsqp_batch *b);
/* Per problem after a round: mode (1 / -1: running and asking; otherwise the exit mode, 0 = success)
 * and the major-iteration count (scipy's nit). */
const int32_t *mpc_slsqp_mode(mpc_slsqp_batch *b);
const int32_t *mpc_slsqp_iterations(mpc_slsqp_batch *b);

/* One round: every unfinished problem makes one core call.  want_f / want_g (each [capacity]) receive the
 * ascending indices of the problems now asking for (f, c) / (grad f, dc/dx).  Returns the number of
 * problems still running (= *n_want_f + *n_want_g), or a negative code. */
int mpc_slsqp_advance(mpc_slsqp_batch *b, int32_t *want_f, int32_t *n_want_f, int32_t *want_g, int32_t *n_want_g);

/* QP offload (slsqp_core.h, slsqp_state.offload): the body of every problem stops where it needs the QP
 * subproblem solved and leaves the LDL' factor of the BFGS matrix, its updates and the QP to the caller
 * (mpc_qp_step of libmpc_amd keeps them on the GPU).  Switch on BEFORE mpc_slsqp_begin.  Then use
 * mpc_slsqp_advance_qp: want_q lists the problems waiting for a QP; for each of them the caller reads
 * lo / hi (rows [n+1], first n entries: the box xl - x, xu - x around the iterate), s (row [n+1], first n: the
 * last step, input of the BFGS update) and qp_flags (bit 0: reset the factor to the identity first, bit 1: update it
 * with the new derivatives first), and writes the new step into s, the m constraint multipliers into r
 * (rows [m + 2n + 2]) and the QP's mode into qp_mode.  In this mode the host needs f, c (mode 1) and grad f
 * (mode -1) only: dc/dx stays with the caller.
 * on = 1: a problem whose linearisation is inconsistent (QP mode 4) ends with mode 104 and is to be solved again
 * without offload.  on = 2: the caller takes Kraft's augmented problem as well: such a problem asks again with
 * qp_flags bit 2 set and the weight's exponent k in bits 3..5 (one more variable in [0, 1], column max(-c, 0),
 * weight 100 * 10^k; no reset, no update); the caller answers as before and puts the extra variable's step into
 * r[m].  (mpc_qp_step_serial, a team of one on the host, serves both kinds of request: tests use it.) */
int mpc_slsqp_set_offload(mpc_slsqp_batch *b, int on);
int mpc_slsqp_advance_qp(mpc_slsqp_batch *b, int32_t *want_f, int32_t *n_want_f, int32_t *want_g, int32_t *n_want_g,
                         int32_t *want_q, int32_t *n_want_q);
double *mpc_slsqp_lo(mpc_slsqp_batch *b);
double *mpc_slsqp_hi(mpc_slsqp_batch *b);
double *mpc_slsqp_s(mpc_slsqp_batch *b);
double *mpc_slsqp_r(mpc_slsqp_batch *b);
int32_t *mpc_slsqp_qp_mode(mpc_slsqp_batch *b);
const int32_t *mpc_slsqp_qp_flags(mpc_slsqp_batch *b);

/* QP offload, staging.  One round's requests (the lists mpc_slsqp_advance_qp returned) move through two flat
 * buffers of the caller's (pinned host memory, one copy each way), filled and emptied row by row on the batch's
 * threads.  With nf / ng / nq requests for f and c / grad f / a QP:
 *   pack    stage  = x[nf][n], x[ng][n] (clipped to the box: the points to evaluate), lo[nq][n], hi[nq][n], s[nq][n]
 *           istage = want_f[nf], want_g[ng], want_q[nq], qp_flags[nq]
 *   unpack  stage  = c[nf][m], grad f[ng][n], step[nq][n], multipliers[nq][m], f[nf]
 *           istage = qp_mode[nq]   (step and multipliers are taken only where the mode is 1)
 * Returns 0, or -1 for bad arguments. */
int mpc_slsqp_pack(mpc_slsqp_batch *b, int nf, const int32_t *want_f, int ng, const int32_t *want_g, int nq,
                   const int32_t *want_q, double *stage, int32_t *istage);
int mpc_slsqp_unpack(mpc_slsqp_batch *b, int nf, const int32_t *want_f, int ng, const int32_t *want_g, int nq,
                     const int32_t *want_q, const double *stage, const int32_t *istage);

/* Seconds spent inside the core by the busiest thread / wall seconds of all rounds since mpc_slsqp_begin. */
double mpc_slsqp_core_seconds(const mpc_slsqp_batch *b);

#ifdef __cplusplus
}
#endif
#endif /* MPC_SLSQP_H */
