// Host instantiation of the team-parallel LSQ (lsq_spmd.hpp) with a team of one, behind the signature of
// slsqp_core.c's lsq: tests install it with mpc_slsqp_set_lsq and require whole solves to stay bit-identical.
#include <cmath>
#include <cstdlib>
#include <vector>

#include "lsq_spmd.hpp"
#include "slsqp_core.h"

// diagnostic: NNLS iterations (columns entered or left) of the last mpc_lsq_spmd_serial call
extern "C" {
int mpc_lsq_spmd_nnls_iterations = 0;
}

// diagnostic: how many of the calls so far were Kraft's augmented problem (tests make sure the case was met)
extern "C" {
int mpc_lsq_spmd_augmented_calls = 0;
int mpc_lsq_spmd_free_steps = 0;          // ... calls that ended with the unconstrained step (lsq's MPC_QP_FREE_STEP)
int mpc_lsq_spmd_free_step_errors = 0;   // ... (a build with MPC_QP_FREE_STEP=2) predictions that did not hold
int mpc_lsq_spmd_plain_calls = 0;         // ... of so many calls of the plain problem
int mpc_lsq_spmd_diagonal_calls = 0;      // ... and calls that took the short forms for a diagonal factor
}

extern "C" void mpc_lsq_spmd_serial(int m, int meq, int n, int nl, int la, double *l, double *g, double *a, double *b,
                                    double *xl, double *xu, double *x, double *y, double *w, int *jw, int *mode)
{
    // the plain problem has n = 6H variables and the whole factor; the augmented one n = 6H + 1 (the last in [0, 1])
    // and the factor of the first 6H with rho behind it
    const bool plain = n % 6 == 0 && nl == (n + 1) * n / 2 + 1;
    const bool aug = n % 6 == 1 && nl == n * (n - 1) / 2 + 1 && xl[n - 1] == 0.0 && xu[n - 1] == 1.0;
    const int nv = aug ? n - 1 : n, H = nv / 6;
    if (meq != 0 || !(plain || aug) || H < 1 || m != 12 * (H + 1)) {
        mpc_slsqp_lsq_builtin(m, meq, n, nl, la, l, g, a, b, xl, xu, x, y, w, jw, mode);
        return;
    }
    const int nzc = 6 * H + 3 * H * (H + 1), mg = m + 2 * nv;
    std::vector<double> nz(nzc), dw(lsqspmd::work_doubles(nv, m, 1, aug ? 1 : 0)), yy(mg);
    std::vector<int> iw(lsqspmd::work_ints(nv, m, 1, aug ? 1 : 0));
    for (int j = 0; j < H; ++j) {
        const int tri = j * H - (j * (j - 1)) / 2 - j - 1;
        for (int q = 0; q < 6; ++q) {
            const int col = 6 * j + q;
            nz[col] = a[(12 * (j + 1) + 6 + q) + (size_t)col * la];
            for (int k = j + 1; k <= H; ++k) nz[6 * H + 6 * (tri + k) + q] = a[(12 * k + q) + (size_t)col * la];
        }
    }
    lsqspmd::Work W = lsqspmd::carve(dw.data(), dw.data() + lsqspmd::big_doubles(nv, m, aug ? 1 : 0), iw.data(), nv, m, 1, aug ? 1 : 0);
    for (int e = 0; e < nl; ++e) W.E[e] = l[e];                  // lsq works on its own copy of the factor
    if (aug) {
        mpc_lsq_spmd_augmented_calls++;
        *mode = lsqspmd::lsq<true>(lsqspmd::SerialTeam(), W, H, g, b, nz.data(), xl, xu, x, yy.data(), &x[nv]);
    } else {
        // a diagonal factor (the first QP of a solve, the one after a reset) takes lsq's short forms, as on the device
        bool diagonal = true;
        for (int r = 0, e = 0; r < nv && diagonal; ++r)
            for (int c = r; c < nv; ++c, ++e)
                if (c != r && (l[e] != 0.0 || std::signbit(l[e]))) { diagonal = false; break; }
        mpc_lsq_spmd_diagonal_calls += diagonal ? 1 : 0;
        W.isc[5] = 0;
        *mode = lsqspmd::lsq(lsqspmd::SerialTeam(), W, H, g, b, nz.data(), xl, xu, x, yy.data(), nullptr, diagonal);
        mpc_lsq_spmd_plain_calls++;
        mpc_lsq_spmd_free_steps += W.isc[5] ? 1 : 0;
        mpc_lsq_spmd_free_step_errors += W.isc[5] == 2 ? 1 : 0;
    }
    mpc_lsq_spmd_nnls_iterations = W.isc[3];
    if (*mode == 1) {
        for (int i = 0; i < m; ++i) y[i] = yy[i];
        for (int i = m; i < m + 2 * nv; ++i) y[i] = NAN;
    }
}

extern "C" void mpc_bfgs_spmd_serial(int m, int n, int la, double *l, double *g, double *a, double *r, double *u, double *v,
                                     double *s)
{
    const int H = n / 6;
    if (n != 6 * H || m != 12 * (H + 1)) {
        mpc_slsqp_bfgs_builtin(m, n, la, l, g, a, r, u, v, s);
        return;
    }
    const int nzc = 6 * H + 3 * H * (H + 1);
    std::vector<double> nz(nzc), dw(lsqspmd::work_doubles(n, m, 1));
    std::vector<int> iw(lsqspmd::work_ints(n, m, 1));
    for (int j = 0; j < H; ++j) {
        const int tri = j * H - (j * (j - 1)) / 2 - j - 1;
        for (int q = 0; q < 6; ++q) {
            const int col = 6 * j + q;
            nz[col] = a[(12 * (j + 1) + 6 + q) + (size_t)col * la];
            for (int k = j + 1; k <= H; ++k) nz[6 * H + 6 * (tri + k) + q] = a[(12 * k + q) + (size_t)col * la];
        }
    }
    lsqspmd::Work W = lsqspmd::carve(dw.data(), dw.data() + lsqspmd::big_doubles(n, m), iw.data(), n, m, 1);
    lsqspmd::bfgs_update(lsqspmd::SerialTeam(), W, H, l, g, nz.data(), r, v, s, u);
}

// One request of mpc_qp_step (kernels_qp.hip: qp_step_kernel) for one problem, with a team of one on the host: reset
// or BFGS-update the factor l, solve the QP (plain or augmented, see the flags in slsqp_core.h), keep the multipliers
// in r [m + 2n + 2] and the Lagrangian gradient in v.  Tests drive the QP-offload protocol of slsqp_body.inc with it
// where there is no GPU.  lo / hi / s: the request's box and the last step (in), the new step (out, if the mode is 1).
extern "C" int mpc_qp_step_serial(int H, int flags, const double *lo, const double *hi, double *s, double *l, const double *g,
                                  const double *c, const double *nz, double *r, double *v)
{
    const int n = 6 * H, m = 12 * (H + 1), ldl = (n + 1) * n / 2 + 1, aug = (flags & MPC_QP_FLAG_AUG) ? 1 : 0;
    std::vector<double> dw(lsqspmd::work_doubles(n, m, 1, aug)), s_in(s, s + n);
    std::vector<int> iw(lsqspmd::work_ints(n, m, 1, aug));
    const lsqspmd::SerialTeam T;
    lsqspmd::Work W = lsqspmd::carve(dw.data(), dw.data() + lsqspmd::big_doubles(n, m, aug), iw.data(), n, m, 1, aug);
    if (flags & MPC_QP_FLAG_RESET) {
        for (int e = 0; e < ldl - 1; ++e) W.E[e] = 0.0;
        for (int i = 0; i < n; ++i) W.E[i * n - (i * (i - 1)) / 2] = 1.0;
    } else {
        for (int e = 0; e < ldl - 1; ++e) W.E[e] = l[e];
    }
    if (flags & MPC_QP_FLAG_BFGS) {
        for (int i = 0; i < n; ++i) { W.bb[i] = v[i]; W.zz[i] = s_in[i]; }
        lsqspmd::bfgs_update(T, W, H, W.E, g, nz, r, W.bb, W.zz, W.xs);
    }
    if (flags & (MPC_QP_FLAG_RESET | MPC_QP_FLAG_BFGS))
        for (int e = 0; e < ldl - 1; ++e) l[e] = W.E[e];
    int md;
    if (flags & MPC_QP_FLAG_AUG) {
        double rho = 100.0;
        for (int k = (flags >> 3) & 7; k > 0; --k) rho = 10.0 * rho;
        W.E[ldl - 1] = rho;
        md = lsqspmd::lsq<true>(T, W, H, g, c, nz, lo, hi, s, r, r + m);
    } else {
        W.isc[5] = 0;
        md = lsqspmd::lsq(T, W, H, g, c, nz, lo, hi, s, r, nullptr, (flags & 3) == 1);
        mpc_lsq_spmd_plain_calls++;
        mpc_lsq_spmd_free_steps += W.isc[5] ? 1 : 0;
        mpc_lsq_spmd_free_step_errors += W.isc[5] == 2 ? 1 : 0;
    }
    if (md == 1)
        for (int i = 0; i < n; ++i) v[i] = g[i] - lsqspmd::jac_column_dot(H, nz, r, i);
    return md;
}
