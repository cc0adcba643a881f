/*
 * TEST INFRASTRUCTURE — plain-C restatement of the reference's MPC inner loop.
 *
 * This is the oracle, not the product: only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg load the library built from this file.
 * Nothing under mpc_amd/ links or calls it.
 *
 * Scalar fp64, sequential evaluation order, compiled with -ffp-contract=off so
 * that every product and sum rounds once, like the numpy expressions it
 * follows.  Citations: MPC:<line> = ur_mpc.py, DYN:<line> = ur20_dynamics.py
 * (both under /root/reference/ur20_mpc_controller/ur20_mpc_controller/models/),
 * NUMDIFF:<line> = scipy/optimize/_numdiff.py (scipy 1.15.3, third-party, the
 * finite-difference scheme SLSQP applies to the reference's callables).
 *
 * Pinned by: tests/test_oracle.py (vs the numpy restatement, which is itself
 * compared bit-for-bit with the real reference when /root/reference exists)
 * and the .npz files in tests/golden/ (vectors produced by the real reference).
 * Jacobian arithmetic (MoveIt in the reference, DYN:82) is this repo's own:
 * "parity unpinned" at that boundary.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct orc_params {
    int32_t horizon;
    int32_t reserved;
    double dt_cost;  /* UR20Dynamics.dt: rollout inside _cost_function (MPC:686, DYN:97-98) */
    double dt_con;   /* URMPC.dt: predict_trajectory inside the constraint (MPC:705) */
    double w_position, w_orientation, w_control, w_smooth, w_damping, w_linear, w_angular;
    double terminal_factor; /* 2.0, MPC:373 */
    double fd_eps;          /* 1e-3, MPC:197 */
    double q_lo[6], q_hi[6]; /* DYN:31-38 */
    double v_lim[6];         /* DYN:40-47; also the variable bounds, MPC:420-430 */
    double dh_a[6], dh_d[6], dh_ca[6], dh_sa[6];
    double base_c, base_s;
    double tool[3];
} orc_params;

/* ---- kinematics: same op order as oracle/numpy_ref.py:_chain ------------ */
static void chain(const orc_params *P, const double q[6], double zs[6][3], double os[6][3],
                  double R[3][3], double pe[3])
{
    double o[3] = {0.0, 0.0, 0.0};
    R[0][0] = P->base_c; R[0][1] = -P->base_s; R[0][2] = 0.0;
    R[1][0] = P->base_s; R[1][1] = P->base_c;  R[1][2] = 0.0;
    R[2][0] = 0.0;       R[2][1] = 0.0;        R[2][2] = 1.0;
    for (int i = 0; i < 6; ++i) {
        for (int r = 0; r < 3; ++r) { zs[i][r] = R[r][2]; os[i][r] = o[r]; }
        const double c = cos(q[i]), s = sin(q[i]);
        const double ca = P->dh_ca[i], sa = P->dh_sa[i], a = P->dh_a[i], d = P->dh_d[i];
        const double tx = a * c, ty = a * s, tz = d;
        const double m01 = -(s * ca), m02 = s * sa, m11 = c * ca, m12 = -(c * sa);
        double Rn[3][3];
        for (int r = 0; r < 3; ++r) {
            o[r] = o[r] + ((R[r][0] * tx + R[r][1] * ty) + R[r][2] * tz);
            Rn[r][0] = R[r][0] * c + R[r][1] * s;
            Rn[r][1] = (R[r][0] * m01 + R[r][1] * m11) + R[r][2] * sa;
            Rn[r][2] = (R[r][0] * m02 + R[r][1] * m12) + R[r][2] * ca;
        }
        memcpy(R, Rn, sizeof(Rn));
    }
    for (int r = 0; r < 3; ++r)
        pe[r] = o[r] + ((R[r][0] * P->tool[0] + R[r][1] * P->tool[1]) + R[r][2] * P->tool[2]);
}

/* a5: stand-in for MoveIt getJacobian (DYN:74-87); J row-major 6x6 */
void orc_jacobian(const orc_params *P, const double q[6], double J[36])
{
    double zs[6][3], os[6][3], R[3][3], pe[3];
    chain(P, q, zs, os, R, pe);
    for (int i = 0; i < 6; ++i) {
        const double z0 = zs[i][0], z1 = zs[i][1], z2 = zs[i][2];
        const double d0 = pe[0] - os[i][0], d1 = pe[1] - os[i][1], d2 = pe[2] - os[i][2];
        J[0 * 6 + i] = z1 * d2 - z2 * d1;
        J[1 * 6 + i] = z2 * d0 - z0 * d2;
        J[2 * 6 + i] = z0 * d1 - z1 * d0;
        J[3 * 6 + i] = z0;
        J[4 * 6 + i] = z1;
        J[5 * 6 + i] = z2;
    }
}

void orc_fk(const orc_params *P, const double q[6], double pos[3], double rpy[3])
{
    double zs[6][3], os[6][3], R[3][3];
    chain(P, q, zs, os, R, pos);
    const double cy = sqrt(R[0][0] * R[0][0] + R[1][0] * R[1][0]);
    if (cy > 1e-12) {
        rpy[0] = atan2(R[2][1], R[2][2]);
        rpy[1] = atan2(-R[2][0], cy);
        rpy[2] = atan2(R[1][0], R[0][0]);
    } else {
        rpy[0] = atan2(-R[1][2], R[1][1]);
        rpy[1] = atan2(-R[2][0], cy);
        rpy[2] = 0.0;
    }
}

/* sequential 6-term dot product, one rounding per op */
static inline double dot6(const double *row, const double *u)
{
    double acc = row[0] * u[0];
    for (int c = 1; c < 6; ++c) acc = acc + row[c] * u[c];
    return acc;
}

static inline double sumsq3(const double e[3]) { return (e[0] * e[0] + e[1] * e[1]) + e[2] * e[2]; }

static inline double sumsq6(const double e[6])
{
    double acc = e[0] * e[0];
    for (int c = 1; c < 6; ++c) acc = acc + e[c] * e[c];
    return acc;
}

/* MPC:263-283 (stage) / MPC:347-362 (terminal): compensation targets in the base frame.
 * b = [P(3), Theta(3), V(3), Omega(3)] */
static void comp_targets(const double p[3], const double *b, double tl[3], double ta[3])
{
    const double *Pp = b, *Th = b + 3, *V = b + 6, *Om = b + 9;
    const double r0 = p[0] - Pp[0], r1 = p[1] - Pp[1], r2 = p[2] - Pp[2];
    /* np.cross(Omega, r) */
    const double x0 = Om[1] * r2 - Om[2] * r1;
    const double x1 = Om[2] * r0 - Om[0] * r2;
    const double x2 = Om[0] * r1 - Om[1] * r0;
    const double w0 = -(V[0] + x0), w1 = -(V[1] + x1), w2 = -(V[2] + x2);
    const double a0 = -Om[0], a1 = -Om[1], a2 = -Om[2];
    const double c = cos(Th[2]), s = sin(Th[2]);
    /* R = [[c,s,0],[-s,c,0],[0,0,1]] @ v   (MPC:714-720) */
    tl[0] = (c * w0 + s * w1) + 0.0 * w2;
    tl[1] = (-s * w0 + c * w1) + 0.0 * w2;
    tl[2] = (0.0 * w0 + 0.0 * w1) + 1.0 * w2;
    ta[0] = (c * a0 + s * a1) + 0.0 * a2;
    ta[1] = (-s * a0 + c * a1) + 0.0 * a2;
    ta[2] = (0.0 * a0 + 0.0 * a1) + 1.0 * a2;
}

/* a1 (+a2,a3,a4): MPC:665-692.  decay[k] = 0.95**k evaluated by the caller
 * (CPython pow) so the table is bit-identical to the reference's MPC:294. */
double orc_cost(const orc_params *P, const double *decay, const double *x, const double *state0,
                const double *target, const double *btraj, const double *Jc, const double *uprev,
                int has_prev)
{
    const int H = P->horizon;
    double q[6], p[3], o[3], J[36];
    for (int j = 0; j < 6; ++j) q[j] = state0[j];
    for (int j = 0; j < 3; ++j) { p[j] = state0[12 + j]; o[j] = state0[15 + j]; }
    double total = 0.0;
    for (int k = 0; k < H; ++k) {
        const double *u = x + 6 * k;
        const double *b = btraj + 12 * k;
        /* ---- stage cost, MPC:249-328 ---- */
        double vl[3], va[3], tl[3], ta[3], e[3];
        for (int r = 0; r < 3; ++r) { vl[r] = dot6(Jc + 6 * r, u); va[r] = dot6(Jc + 6 * (3 + r), u); }
        comp_targets(p, b, tl, ta);
        const double dk = decay[k];
        for (int r = 0; r < 3; ++r) e[r] = vl[r] - tl[r];
        const double c_lin = P->w_linear * dk * sumsq3(e);
        for (int r = 0; r < 3; ++r) e[r] = va[r] - ta[r];
        const double c_ang = P->w_angular * dk * sumsq3(e);
        for (int r = 0; r < 3; ++r) e[r] = p[r] - target[r];
        const double c_pos = P->w_position * dk * sumsq3(e);
        for (int r = 0; r < 3; ++r) e[r] = o[r] - target[3 + r];
        const double c_ori = P->w_orientation * dk * sumsq3(e);
        const double uu = sumsq6(u);
        const double c_ctl = P->w_control * uu;
        double c_damp = 0.0, c_smooth = 0.0;
        if (k > 0 && has_prev) {
            double du[6];
            for (int j = 0; j < 6; ++j) du[j] = u[j] - uprev[j];
            c_damp = P->w_damping * sumsq6(du);
        }
        if (k > 0 && H > 1) c_smooth = P->w_smooth * uu;
        const double stage = (((((c_lin + c_ang) + c_pos) + c_ori) + c_ctl) + c_damp) + c_smooth;
        total = total + stage;
        /* ---- Euler step, DYN:101-128 ---- */
        orc_jacobian(P, q, J);
        double w[6];
        for (int r = 0; r < 6; ++r) w[r] = dot6(J + 6 * r, u);
        for (int j = 0; j < 6; ++j) q[j] = q[j] + u[j] * P->dt_cost;
        for (int r = 0; r < 3; ++r) { p[r] = p[r] + w[r] * P->dt_cost; o[r] = o[r] + w[3 + r] * P->dt_cost; }
    }
    /* ---- terminal cost, MPC:330-381; dq_H = u_{H-1}, base = B_{H-1} (MPC:689) ---- */
    {
        const double *dq = x + 6 * (H - 1);
        const double *b = btraj + 12 * (H - 1);
        double vl[3], va[3], tl[3], ta[3], e[3];
        orc_jacobian(P, q, J);
        for (int r = 0; r < 3; ++r) { vl[r] = dot6(J + 6 * r, dq); va[r] = dot6(J + 6 * (3 + r), dq); }
        comp_targets(p, b, tl, ta);
        for (int r = 0; r < 3; ++r) e[r] = vl[r] - tl[r];
        const double c_lin = P->w_linear * P->terminal_factor * sumsq3(e);
        for (int r = 0; r < 3; ++r) e[r] = va[r] - ta[r];
        const double c_ang = P->w_angular * P->terminal_factor * sumsq3(e);
        for (int r = 0; r < 3; ++r) e[r] = p[r] - target[r];
        const double c_pos = P->w_position * sumsq3(e);
        for (int r = 0; r < 3; ++r) e[r] = o[r] - target[3 + r];
        const double c_ori = P->w_orientation * sumsq3(e);
        total = total + (((c_lin + c_ang) + c_pos) + c_ori);
    }
    return total;
}

/* a6: MPC:388-411; c has 12*(H+1) rows, state 0 included */
void orc_constraint(const orc_params *P, const double *x, const double *state0, double *c)
{
    const int H = P->horizon;
    double q[6], dq[6];
    for (int j = 0; j < 6; ++j) { q[j] = state0[j]; dq[j] = state0[6 + j]; }
    for (int k = 0; k <= H; ++k) {
        for (int j = 0; j < 6; ++j) {
            const double lo = q[j] - P->q_lo[j], hi = P->q_hi[j] - q[j];
            c[12 * k + j] = fmin(lo, hi); /* np.minimum on finite values */
            c[12 * k + 6 + j] = P->v_lim[j] - fabs(dq[j]);
        }
        if (k < H)
            for (int j = 0; j < 6; ++j) { q[j] = q[j] + x[6 * k + j] * P->dt_con; dq[j] = x[6 * k + j]; }
    }
}

/* NUMDIFF:498-524 with abs_step, then _adjust_scheme_to_bounds '1-sided' (NUMDIFF:44-72) */
double orc_fd_step(double x0, double lb, double ub, double eps)
{
    double h = eps;
    const double dx = (x0 + h) - x0;
    if (dx == 0.0) {
        const double sgn = (x0 >= 0.0) ? 1.0 : -1.0;
        h = sqrt(DBL_EPSILON) * sgn * fmax(1.0, fabs(x0));
    }
    const double lower = x0 - lb, upper = ub - x0;
    const double xs = x0 + h;
    const int violated = (xs < lb) || (xs > ub);
    const int fitting = fabs(h) <= fmax(lower, upper);
    if (violated && fitting) h = -h;
    if (!fitting) h = (upper >= lower) ? upper : -lower;
    return h;
}

/* a10: one "eval" = f, c, grad f (n), dc/dx.  jac_cm is column-major m x n
 * (jac_cm[i*m + r] = d c_r / d x_i), the layout SLSQP's Fortran core consumes. */
void orc_eval_fd(const orc_params *P, const double *decay, const double *x, const double *state0,
                 const double *target, const double *btraj, const double *Jc, const double *uprev,
                 int has_prev, double *f_out, double *grad, double *c, double *jac_cm)
{
    const int H = P->horizon, n = 6 * H, m = 12 * (H + 1);
    double *x1 = (double *)malloc(sizeof(double) * (size_t)n);
    double *c1 = (double *)malloc(sizeof(double) * (size_t)m);
    memcpy(x1, x, sizeof(double) * (size_t)n);
    const double f0 = orc_cost(P, decay, x, state0, target, btraj, Jc, uprev, has_prev);
    orc_constraint(P, x, state0, c);
    *f_out = f0;
    for (int i = 0; i < n; ++i) {
        const int j = i % 6;
        const double h = orc_fd_step(x[i], -P->v_lim[j], P->v_lim[j], P->fd_eps);
        x1[i] = x[i] + h;
        const double dx = x1[i] - x[i]; /* NUMDIFF:596 */
        if (grad) {
            const double f1 = orc_cost(P, decay, x1, state0, target, btraj, Jc, uprev, has_prev);
            grad[i] = (f1 - f0) / dx;
        }
        if (jac_cm) {
            orc_constraint(P, x1, state0, c1);
            for (int r = 0; r < m; ++r) jac_cm[(size_t)i * m + r] = (c1[r] - c[r]) / dx;
        }
        x1[i] = x[i];
    }
    free(x1);
    free(c1);
}

/* batch driver over scenarios laid out [N][...]; any output pointer may be NULL */
void orc_eval_fd_batch(const orc_params *P, const double *decay, int N, const double *x,
                       const double *state0, const double *target, const double *btraj,
                       const double *Jc, const double *uprev, const uint8_t *has_prev, double *cost,
                       double *grad, double *con, double *jac_cm)
{
    const int H = P->horizon, n = 6 * H, m = 12 * (H + 1);
    double *ctmp = (double *)malloc(sizeof(double) * (size_t)m);
    for (int s = 0; s < N; ++s) {
        double f;
        orc_eval_fd(P, decay, x + (size_t)s * n, state0 + (size_t)s * 18, target + (size_t)s * 6,
                    btraj + (size_t)s * 12 * H, Jc + (size_t)s * 36, uprev ? uprev + (size_t)s * 6 : NULL,
                    has_prev ? has_prev[s] : 0, &f, grad ? grad + (size_t)s * n : NULL,
                    con ? con + (size_t)s * m : ctmp, jac_cm ? jac_cm + (size_t)s * n * m : NULL);
        if (cost) cost[s] = f;
    }
    free(ctmp);
}

/* f and c only (SLSQP mode 1, _slsqp_py.py:434-436) */
void orc_eval_batch(const orc_params *P, const double *decay, int N, const double *x,
                    const double *state0, const double *target, const double *btraj, const double *Jc,
                    const double *uprev, const uint8_t *has_prev, double *cost, double *con)
{
    const int H = P->horizon, n = 6 * H, m = 12 * (H + 1);
    for (int s = 0; s < N; ++s) {
        cost[s] = orc_cost(P, decay, x + (size_t)s * n, state0 + (size_t)s * 18, target + (size_t)s * 6,
                           btraj + (size_t)s * 12 * H, Jc + (size_t)s * 36,
                           uprev ? uprev + (size_t)s * 6 : NULL, has_prev ? has_prev[s] : 0);
        if (con) orc_constraint(P, x + (size_t)s * n, state0 + (size_t)s * 18, con + (size_t)s * m);
    }
}

/* a12: _predict_base_dynamic, MPC:606-651.  base = [P(3),Theta(3),V(3),Omega(3)],
 * dyn = [mass, I_z, k_v, k_omega, tau_linear, tau_angular]; v_cmd/omega_cmd default to v/omega */
void orc_predict_base_dynamic(const double *base, const double *dyn, double dt, int H, double *out)
{
    double pos[3] = {base[0], base[1], base[2]};
    double yaw = base[5], v = base[6], w = base[11];
    const double v_cmd = v, w_cmd = w;
    for (int k = 0; k < H; ++k) {
        const double force = (dyn[0] / dyn[4]) * (v_cmd - v);
        const double acc = (force - dyn[2] * v) / dyn[0];
        v = v + acc * dt;
        const double torque = (dyn[1] / dyn[5]) * (w_cmd - w);
        const double alpha = (torque - dyn[3] * w) / dyn[1];
        w = w + alpha * dt;
        yaw = yaw + w * dt;
        const double c = cos(yaw), s = sin(yaw);
        const double vx = v * c, vy = v * s;
        pos[0] = pos[0] + vx * dt; pos[1] = pos[1] + vy * dt; pos[2] = pos[2] + 0.0 * dt;
        double *r = out + 12 * k;
        r[0] = pos[0]; r[1] = pos[1]; r[2] = pos[2];
        r[3] = 0.0; r[4] = 0.0; r[5] = yaw;
        r[6] = vx; r[7] = vy; r[8] = 0.0;
        r[9] = 0.0; r[10] = 0.0; r[11] = w;
    }
}
