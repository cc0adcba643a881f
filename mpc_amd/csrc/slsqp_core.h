/* SLSQP core (host C): one reverse-communication call, see slsqp_core.c. */
#ifndef MPC_SLSQP_CORE_H
#define MPC_SLSQP_CORE_H

#ifdef __cplusplus
extern "C" {
#endif

/* the scalars the Fortran original keeps in SAVEd variables / scipy passes back and forth (_slsqp_py.py:396-409) */
typedef struct slsqp_state {
    double alpha, f0, gs, h1, h2, h3, h4, t, t0, tol;
    int iexact, incons, ireset, itermx, line, n1, n2, n3;
    /* QP offload (all zero = everything on the host).  With `offload` set the body never touches l or a: where
     * it would reset l it sets qp_reset, where it would update l (BFGS) it sets qp_bfgs, and where it would solve
     * the QP it returns MPC_SLSQP_MODE_QP with u = xl - x and v = xu - x in place.  The caller resets / updates l,
     * solves the QP (mpc_qp_step in include/mpc_amd.h does all three on the device), writes the step into s, the
     * constraint multipliers into r, the QP's mode into qp_mode, and calls again with mode = MPC_SLSQP_MODE_QP.
     * offload = 2: the caller also solves Kraft's augmented problem (inconsistent linearisation, QP mode 4): the body
     * asks again with qp_aug set (no reset, no update; the weight of the extra variable is 100 * 10^incons, Kraft's
     * l(n3)), the caller answers with the step's extra entry in qp_delta as well, up to six times with ten times the
     * weight.  offload = 1: such a solve ends with MPC_SLSQP_MODE_NEEDS_HOST. */
    int offload, qp_mode, qp_reset, qp_bfgs;
    int qp_aug, qp_unused;
    double qp_delta;
} slsqp_state;

#define MPC_SLSQP_MODE_QP 20
/* exit mode of a solve offloaded to a caller that does not take Kraft's augmented problem (offload = 1) whose
 * linearisation was inconsistent: the caller solves that problem again with the host-only driver */
#define MPC_SLSQP_MODE_NEEDS_HOST 104
/* the QP request's flags (mpc_qp_step): reset l / BFGS-update l / the augmented problem, its weight 100 * 10^k in
 * bits 3..5 */
#define MPC_QP_FLAG_RESET 1
#define MPC_QP_FLAG_BFGS 2
#define MPC_QP_FLAG_AUG 4
#define MPC_QP_FLAGS_OF(S) (((S)->qp_reset ? MPC_QP_FLAG_RESET : 0) | ((S)->qp_bfgs ? MPC_QP_FLAG_BFGS : 0) | \
                            ((S)->qp_aug ? (MPC_QP_FLAG_AUG | (((S)->incons & 7) << 3)) : 0))

/* Body of the method with its persistent vectors passed one by one: r [m + 2n + 2] multipliers,
 * l [(n+1)n/2 + 1] LDL' factor of the BFGS matrix, x0 [n], mu [m], s / u / v [n+1], and per-call
 * scratch w (doubles) / iw (ints).  a is column-major (la, n+1).  mode: 0 on the first call, then
 * what the caller was asked for (1: f and c are fresh, -1: g and a are fresh). */
void mpc_slsqpb(int m, int meq, int la, int n, double *x, double *xl, double *xu, double f, double *c, double *g,
                double *a, double *acc, int *iter, int *mode, double *r, double *l, double *x0, double *mu, double *s,
                double *u, double *v, double *w, int *iw, slsqp_state *S);

/* The QP subproblem alone (Kraft's LSQ), and a hook that replaces it inside mpc_slsqpb: the way the
 * team-parallel restatement (lsq_spmd.hpp, the form a GPU workgroup runs) is checked against this file
 * over whole solves.  NULL restores the built-in routine.  Process-wide; meant for tests. */
typedef void (*mpc_lsq_fn)(int m, int meq, int n, int nl, int la, double *l, double *g, double *a, double *b, double *xl,
                           double *xu, double *x, double *y, double *w, int *jw, int *mode);
void mpc_slsqp_lsq_builtin(int m, int meq, int n, int nl, int la, double *l, double *g, double *a, double *b, double *xl,
                           double *xu, double *x, double *y, double *w, int *jw, int *mode);
void mpc_slsqp_set_lsq(mpc_lsq_fn fn);
/* Likewise for the BFGS update block of the major iteration (l is updated in place; u, v are the body's work
 * vectors, v holding the previous Lagrangian gradient on entry). */
typedef void (*mpc_bfgs_fn)(int m, int n, int la, double *l, double *g, double *a, double *r, double *u, double *v, double *s);
void mpc_slsqp_bfgs_builtin(int m, int n, int la, double *l, double *g, double *a, double *r, double *u, double *v, double *s);
void mpc_slsqp_set_bfgs(mpc_bfgs_fn fn);

/* Same call with scipy's single workspace array (layout of the Fortran driver SLSQP), for side-by-side tests. */
void mpc_slsqp_step(int m, int meq, int la, int n, double *x, double *xl, double *xu, double f, double *c, double *g,
                    double *a, double *acc, int *iter, int *mode, double *w, int l_w, int *jw, int l_jw, slsqp_state *S);

#ifdef __cplusplus
}
#endif
#endif
