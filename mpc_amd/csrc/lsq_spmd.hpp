// The QP subproblem of SLSQP (Kraft's LSQ: LSI -> LDP -> NNLS, Lawson & Hanson) written for a TEAM of
// threads working on ONE problem: the form a GPU workgroup runs (one workgroup per problem), and, with a
// team of one, plain serial code that is compared with slsqp_core.c on the CPU
// (tests/test_native_slsqp.py).
//
// Rules that keep it bit-identical to slsqp_core.c whatever the team size:
//   * a PAR_FOR body touches only what its own index owns, and computes every entry with the same
//     operations in the same order as the serial text (sums run sequentially INSIDE one index);
//   * everything else (recurrences, the bookkeeping of NNLS) runs on thread 0 between barriers;
//   * scalars every thread needs after a serial section travel through W.sc / W.isc.
// Layout: g is column-major with mg rows and h stored as one more "column", i.e. [g h]^T row by row: the
// matrix NNLS works on in LDP is a straight copy of it (no transposition), and a thread that owns one
// constraint walks memory that its neighbours walk next to it.
//
// Shape is the MPC's: n = 6H variables, m = 12(H+1) constraint rows given in structured form
// (include/mpc_amd.h, mpc_eval_fd_nz), finite box bounds, no equality constraints, no augmented problem
// (an inconsistent linearisation returns mode 4 and the caller falls back to the host core).
#pragma once

#include <math.h>

#if defined(__HIPCC__)
#define SPMD_HD __host__ __device__
#else
#define SPMD_HD
#endif

namespace lsqspmd {

struct SerialTeam {
    SPMD_HD int tid() const { return 0; }
    SPMD_HD int size() const { return 1; }
    SPMD_HD void sync() const {}
};

#define PAR_FOR(T, i, lo, hi) for (int i = (lo) + (T).tid(); i < (hi); i += (T).size())

// Workspace of one team (doubles unless noted).  mg = m + 2n, n1 = n + 1.
struct Work {
    double *E;     // [n*n]   e = sqrt(d) l^T, column-major, leading dimension n
    double *ff;    // [n]
    double *G;     // [(n+1)*mg]: G[i + j*mg] = g(i, j) for j < n; G[i + n*mg] = h(i)
    double *AT;    // [(n+1)*mg]: the copy NNLS works on and overwrites (LDP needs g and h again afterwards)
    double *acc;   // [mg]
    double *bb;    // [n1]  NNLS right-hand side
    double *zz;    // [n1]
    double *xn;    // [mg]  NNLS solution (dual variables of LDP)
    double *wd;    // [mg]  NNLS dual vector
    double *piv;   // [n1]  copy of the pivot column
    double *xs;    // [n]
    double *sc;    // [16]  scalars broadcast from thread 0
    double *redv;  // [team size]
    int *redi;     // [team size]
    int *idx;      // [mg]
    int *isc;      // [16]
};

SPMD_HD inline size_t work_doubles(int n, int m, int team)
{
    const size_t mg = (size_t)m + 2 * (size_t)n, n1 = (size_t)n + 1;
    return (size_t)n * n + n + 2 * n1 * mg + mg + n1 + n1 + mg + mg + n1 + n + 16 + (size_t)team;
}
SPMD_HD inline size_t work_ints(int n, int m, int team) { return (size_t)team + (size_t)m + 2 * (size_t)n + 16; }

SPMD_HD inline Work carve(double *d, int *i, int n, int m, int team)
{
    const size_t mg = (size_t)m + 2 * (size_t)n, n1 = (size_t)n + 1;
    Work W;
    W.E = d;            d += (size_t)n * n;
    W.ff = d;           d += n;
    W.G = d;            d += n1 * mg;
    W.AT = d;           d += n1 * mg;
    W.acc = d;          d += mg;
    W.bb = d;           d += n1;
    W.zz = d;           d += n1;
    W.xn = d;           d += mg;
    W.wd = d;           d += mg;
    W.piv = d;          d += n1;
    W.xs = d;           d += n;
    W.sc = d;           d += 16;
    W.redv = d;
    W.redi = i;         i += team;
    W.idx = i;          i += mg;
    W.isc = i;
    return W;
}

// scaled 2-norm, the arithmetic of slsqp_core.c's dnrm2_
SPMD_HD inline double nrm2(int n, const double *dx, int incx)
{
    double scale = 0.0, ssq = 1.0;
    if (n < 1) return 0.0;
    for (int i = 0, ix = 0; i < n; ++i, ix += incx) {
        if (dx[ix] != 0.0) {
            const double ax = fabs(dx[ix]);
            if (scale < ax) { ssq = 1.0 + ssq * (scale / ax) * (scale / ax); scale = ax; }
            else ssq = ssq + (ax / scale) * (ax / scale);
        }
    }
    return scale * sqrt(ssq);
}

// NNLS on the transposed storage: entry (row r, column j) of Lawson & Hanson's a is AT[j + r*mg]
// (1-based r in 1..mrow, j in 1..ncol).  b = W.bb, x = W.xn, w = W.wd, z = W.zz, index = W.idx.
template <class Team>
SPMD_HD int nnls(const Team &T, const Work &W, int mrow, int ncol, int mg, double *rnorm_out)
{
#define AT(r, j) W.AT[((j) - 1) + (size_t)((r) - 1) * mg]
    const double factor = 1.0e-2;
    double *b = W.bb, *x = W.xn, *w = W.wd, *z = W.zz;
    int *index = W.idx;
    int iter = 0, iz1 = 1, nsetp = 0, npp1 = 1, mode = 1;
    const int itmax = 3 * ncol, iz2 = ncol, m = mrow;
    if (mrow <= 0 || ncol <= 0) return 2;
    PAR_FOR(T, i, 0, ncol) { index[i] = i + 1; x[i] = 0.0; }
    T.sync();
    for (;;) {
        // ---- L110: dual variables of the columns still in Z
        if (iz1 > iz2 || nsetp >= m) break;
        PAR_FOR(T, t, iz1 - 1, iz2) {
            const int j = index[t];
            double s = 0.0;
            for (int r = npp1; r <= m; ++r) s = s + AT(r, j) * b[r - 1];
            w[j - 1] = s;
        }
        T.sync();
        int accepted = 0, j = 0, iz = 0;
        double up = 0.0;
        for (;;) {
            // ---- L130: the largest positive dual, first position wins a tie
            {
                double best = 0.0;
                int besti = 0;
                PAR_FOR(T, t, iz1 - 1, iz2) {
                    const double v = w[index[t] - 1];
                    if (v > best) { best = v; besti = t + 1; }
                }
                W.redv[T.tid()] = best;
                W.redi[T.tid()] = besti;
                T.sync();
                if (T.tid() == 0) {
                    double wmax = 0.0;
                    int izmax = 0;
                    for (int q = 0; q < T.size(); ++q) {
                        const double v = W.redv[q];
                        const int p = W.redi[q];
                        if (p != 0 && (v > wmax || (v == wmax && p < izmax))) { wmax = v; izmax = p; }
                    }
                    W.sc[0] = wmax;
                    W.isc[0] = izmax;
                }
                T.sync();
            }
            if (W.sc[0] <= 0.0) { accepted = -1; break; }      // exit loop A
            iz = W.isc[0];
            j = index[iz - 1];
            // ---- step four: test column j for linear dependence (thread 0)
            if (T.tid() == 0) {
                const double asave = AT(npp1, j);
                // h12 construct on rows npp1..m of column j
                double cl = fabs(AT(npp1, j)), upv = 0.0;
                int ok = 1;
                if (npp1 + 1 <= m) {               // h12 does nothing when l1 > m
                    for (int r = npp1 + 1; r <= m; ++r) { const double sm = fabs(AT(r, j)); cl = sm > cl ? sm : cl; }
                    if (cl > 0.0) {
                        const double clinv = 1.0 / cl;
                        double sm = (AT(npp1, j) * clinv) * (AT(npp1, j) * clinv);
                        for (int r = npp1 + 1; r <= m; ++r) sm = sm + (AT(r, j) * clinv) * (AT(r, j) * clinv);
                        cl = cl * sqrt(sm);
                        if (AT(npp1, j) > 0.0) cl = -cl;
                        upv = AT(npp1, j) - cl;
                        AT(npp1, j) = cl;
                    }
                }
                {
                    // unorm over the rows already in the triangle (stride mg in this storage)
                    const double unorm = nrm2(nsetp, &AT(1, j), mg);
                    const double t = factor * fabs(AT(npp1, j));
                    volatile double d = unorm + t;
                    if (d - unorm <= 0.0) ok = 0;
                }
                if (ok) {
                    for (int r = 0; r < m; ++r) z[r] = b[r];
                    // h12 apply (mode 2) to z with pivot npp1, l1 = npp1 + 1
                    if (npp1 + 1 <= m && fabs(AT(npp1, j)) > 0.0) {
                        double bq = upv * AT(npp1, j);
                        if (bq < 0.0) {
                            bq = 1.0 / bq;
                            double sm = z[npp1 - 1] * upv;
                            for (int r = npp1 + 1; r <= m; ++r) sm = sm + z[r - 1] * AT(r, j);
                            if (sm != 0.0) {
                                sm = sm * bq;
                                z[npp1 - 1] = z[npp1 - 1] + sm * upv;
                                for (int r = npp1 + 1; r <= m; ++r) z[r - 1] = z[r - 1] + sm * AT(r, j);
                            }
                        }
                    }
                    if (!(z[npp1 - 1] / AT(npp1, j) > 0.0)) ok = 0;
                }
                if (!ok) {
                    AT(npp1, j) = asave;
                    w[j - 1] = 0.0;
                }
                W.isc[1] = ok;
                W.sc[1] = upv;
            }
            T.sync();
            if (W.isc[1]) { accepted = 1; up = W.sc[1]; break; }
        }
        if (accepted < 0) break;
        // ---- step five: column j enters the set
        PAR_FOR(T, r, 0, m) { b[r] = z[r]; W.piv[r] = AT(r + 1, j); }
        T.sync();
        if (T.tid() == 0) {
            index[iz - 1] = index[iz1 - 1];
            index[iz1 - 1] = j;
        }
        iz1 = iz1 + 1;
        nsetp = npp1;
        npp1 = npp1 + 1;
        T.sync();
        {
            // h12 apply (pivot nsetp, l1 = npp1) to every column still in Z, one column per index
            const double pv = W.piv[nsetp - 1];
            const int applies = (npp1 <= m) && (fabs(pv) > 0.0);
            double bq = up * pv;
            if (applies && bq < 0.0) {
                bq = 1.0 / bq;
                PAR_FOR(T, t, iz1 - 1, iz2) {
                    const int jj = index[t];
                    double sm = AT(nsetp, jj) * up;
                    for (int r = npp1; r <= m; ++r) sm = sm + AT(r, jj) * W.piv[r - 1];
                    if (sm != 0.0) {
                        sm = sm * bq;
                        AT(nsetp, jj) = AT(nsetp, jj) + sm * up;
                        for (int r = npp1; r <= m; ++r) AT(r, jj) = AT(r, jj) + sm * W.piv[r - 1];
                    }
                }
            }
        }
        PAR_FOR(T, r, npp1, m + 1) AT(r, j) = 0.0;         // rows npp1..m of the new column
        if (T.tid() == 0) w[j - 1] = 0.0;
        T.sync();
        // ---- steps six to eleven (thread 0): solve, step length, possibly drop columns
        if (T.tid() == 0) {
            int status = 0;          // 0: back to L110, 3: iteration limit / empty set
            int jj = 0;
            for (;;) {
                // L180: solve the triangular system for z
                for (int ip = nsetp; ip >= 1; --ip) {
                    if (ip != nsetp) {
                        const double zf = -z[ip];
                        if (zf != 0.0)
                            for (int r = 1; r <= ip; ++r) z[r - 1] = z[r - 1] + zf * AT(r, jj);
                    }
                    jj = index[ip - 1];
                    z[ip - 1] = z[ip - 1] / AT(ip, jj);
                }
                iter = iter + 1;
                if (iter > itmax) { status = 3; break; }
                double alpha = 1.0;
                jj = 0;
                for (int ip = 1; ip <= nsetp; ++ip) {
                    if (z[ip - 1] > 0.0) continue;
                    const int l = index[ip - 1];
                    const double t = -x[l - 1] / (z[ip - 1] - x[l - 1]);
                    if (alpha < t) continue;
                    alpha = t;
                    jj = ip;
                }
                for (int ip = 1; ip <= nsetp; ++ip) {
                    const int l = index[ip - 1];
                    x[l - 1] = (1.0 - alpha) * x[l - 1] + alpha * z[ip - 1];
                }
                if (jj == 0) break;                       // back to L110
                int i = index[jj - 1];
                int emptied = 0;
                for (;;) {
                    // L250: move column i out of the set, restore the triangle with Givens rotations
                    x[i - 1] = 0.0;
                    jj = jj + 1;
                    for (int jc = jj; jc <= nsetp; ++jc) {
                        const int ii = index[jc - 1];
                        index[jc - 2] = ii;
                        double da = AT(jc - 1, ii), db = AT(jc, ii), c, s;
                        {
                            double roe = db, r, zq;
                            if (fabs(da) > fabs(db)) roe = da;
                            const double scale = fabs(da) + fabs(db);
                            if (scale == 0.0) { c = 1.0; s = 0.0; r = 0.0; }
                            else {
                                r = scale * sqrt((da / scale) * (da / scale) + (db / scale) * (db / scale));
                                r = (roe >= 0.0 ? 1.0 : -1.0) * r;
                                c = da / r;
                                s = db / r;
                            }
                            zq = s;
                            if (fabs(c) > 0.0 && fabs(c) <= s) zq = 1.0 / c;
                            da = r;
                            db = zq;
                        }
                        AT(jc - 1, ii) = da;
                        AT(jc, ii) = db;
                        const double t = AT(jc - 1, ii);
                        for (int q = 1; q <= ncol; ++q) {   // rows jc-1 and jc of every column
                            const double dx = AT(jc - 1, q), dy = AT(jc, q);
                            const double tt = c * dx + s * dy;
                            AT(jc, q) = c * dy - s * dx;
                            AT(jc - 1, q) = tt;
                        }
                        AT(jc - 1, ii) = t;
                        AT(jc, ii) = 0.0;
                        {
                            const double dx = b[jc - 2], dy = b[jc - 1];
                            const double tt = c * dx + s * dy;
                            b[jc - 1] = c * dy - s * dx;
                            b[jc - 2] = tt;
                        }
                    }
                    npp1 = nsetp;
                    nsetp = nsetp - 1;
                    iz1 = iz1 - 1;
                    index[iz1 - 1] = i;
                    if (nsetp <= 0) { emptied = 1; break; }
                    int again = 0;
                    for (jj = 1; jj <= nsetp; ++jj) {
                        i = index[jj - 1];
                        if (x[i - 1] <= 0.0) { again = 1; break; }
                    }
                    if (!again) break;
                }
                if (emptied) { status = 3; break; }
                for (int r = 0; r < m; ++r) z[r] = b[r];
            }
            W.isc[2] = status;
            W.isc[3] = iter;
            W.isc[4] = iz1;
            W.isc[5] = nsetp;
            W.isc[6] = npp1;
        }
        T.sync();
        iter = W.isc[3]; iz1 = W.isc[4]; nsetp = W.isc[5]; npp1 = W.isc[6];
        if (W.isc[2] == 3) { mode = 3; break; }
    }
    // ---- L280
    if (T.tid() == 0) {
        const int k = npp1 < m ? npp1 : m;
        *rnorm_out = nrm2(m - nsetp, &b[k - 1], 1);
    }
    if (npp1 > m) PAR_FOR(T, i, 0, ncol) w[i] = 0.0;
    T.sync();
    return mode;
#undef AT
}

// Kraft's LSQ for the MPC shape.  l: packed LDL' factor ((n+1)n/2 + 1 entries), g: gradient, cvals: c(x),
// nz: structured dc/dx, lo / hi: bounds on the step (xl - x, xu - x).  Returns the mode (1 = solved) and
// writes the step s [n] and the multipliers y [m + 2n] (the bound rows' multipliers follow the m
// constraint rows; slsqp_core.c overwrites those with NaN, they are unused).
template <class Team>
SPMD_HD int lsq(const Team &T, const Work &W, int H, const double *l, const double *g, const double *cvals, const double *nz,
                const double *lo, const double *hi, double *s, double *y)
{
    const int n = 6 * H, m = 12 * (H + 1), mg = m + 2 * n;
#define E(i, j) W.E[((i) - 1) + (size_t)((j) - 1) * n]
#define G(i, j) W.G[((i) - 1) + (size_t)((j) - 1) * mg]
    // ---- e = sqrt(d) l^T (upper triangular) and g with the constraint rows, +I, -I
    PAR_FOR(T, e, 0, n * n) {
        const int r = e % n, c = e / n;            // 0-based row, column
        double v = 0.0;
        if (r <= c) {
            const int off = r * n - (r * (r - 1)) / 2;
            const double diag = sqrt(l[off]);
            v = (r == c) ? diag : diag * l[off + (c - r)];
        }
        W.E[e] = v;
    }
    PAR_FOR(T, e, 0, n * mg) W.G[e] = 0.0;
    T.sync();
    PAR_FOR(T, k, 0, 6 * H + 3 * H * (H + 1)) {
        int row, col;
        if (k < 6 * H) {                            // velocity row of state j+1
            col = k;
            row = 12 * (k / 6 + 1) + 6 + k % 6;
        } else {                                    // position rows: k = 6H + 6*T(j,kk) + a
            const int q = (k - 6 * H) / 6, a = (k - 6 * H) % 6;
            int j = 0, base = 0;                    // T(j, kk) = base_j + (kk - j - 1), base_j = j*H - j(j-1)/2
            while (j + 1 < H && (j + 1) * H - ((j + 1) * j) / 2 <= q) ++j;
            base = j * H - (j * (j - 1)) / 2;
            const int kk = q - base + j + 1;
            col = 6 * j + a;
            row = 12 * kk + a;
        }
        W.G[row + (size_t)col * mg] = nz[k];
    }
    PAR_FOR(T, i, 0, m) W.G[i + (size_t)n * mg] = -cvals[i];                 // h = -b
    PAR_FOR(T, v, 0, n) {
        W.G[(m + v) + (size_t)v * mg] = 1.0;
        W.G[(m + n + v) + (size_t)v * mg] = -1.0;
        W.G[(m + v) + (size_t)n * mg] = lo[v];
        W.G[(m + n + v) + (size_t)n * mg] = -hi[v];
    }
    // ---- f: solve e^T f = g by forward substitution, then negate (thread 0: a recurrence)
    T.sync();
    if (T.tid() == 0) {
        for (int i = 1; i <= n; ++i) {
            double t = 0.0;
            for (int k = 1; k < i; ++k) t = t + E(k, i) * W.ff[k - 1];
            W.ff[i - 1] = (g[i - 1] - t) / E(i, i);
        }
        for (int i = 0; i < n; ++i) W.ff[i] = -1.0 * W.ff[i];
        // ---- LSI: "QR" of the triangular e (pivot arithmetic only, see slsqp_core.c) applied to f
        for (int i = 1; i <= n; ++i) {
            double cl = fabs(E(i, i));
            if (i + 1 > n || cl <= 0.0) continue;
            const double clinv = 1.0 / cl;
            double sm = (E(i, i) * clinv) * (E(i, i) * clinv);
            cl = cl * sqrt(sm);
            if (E(i, i) > 0.0) cl = -cl;
            const double up = E(i, i) - cl;
            E(i, i) = cl;
            double bq = up * E(i, i);
            if (bq >= 0.0) continue;
            bq = 1.0 / bq;
            for (int j = i + 1; j <= n; ++j) {
                sm = E(i, j) * up;
                if (sm == 0.0) continue;
                sm = sm * bq;
                E(i, j) = E(i, j) + sm * up;
            }
            sm = W.ff[i - 1] * up;
            if (sm != 0.0) {
                sm = sm * bq;
                W.ff[i - 1] = W.ff[i - 1] + sm * up;
            }
        }
        int regular = 1;
        for (int j = 1; j <= n; ++j)
            if (!(fabs(E(j, j)) >= 2.22e-16)) regular = 0;
        W.isc[8] = regular;
    }
    T.sync();
    if (!W.isc[8]) return 5;
    // ---- g <- g e^-1, h <- h - g f: one row per index, sums in the order of the serial text
    PAR_FOR(T, i, 0, mg) {
        double *gi = W.G + i;
        for (int j = 1; j <= n; ++j) {
            double t = 0.0;
            for (int k = 1; k < j; ++k) t = t + gi[(size_t)(k - 1) * mg] * E(k, j);
            gi[(size_t)(j - 1) * mg] = (gi[(size_t)(j - 1) * mg] - t) / E(j, j);
        }
        double t = 0.0;
        for (int j = 1; j <= n; ++j) t = t + gi[(size_t)(j - 1) * mg] * W.ff[j - 1];
        gi[(size_t)n * mg] = gi[(size_t)n * mg] - t;
    }
    // ---- LDP: NNLS on a copy of [g h]^T with right-hand side e_{n+1}
    T.sync();
    PAR_FOR(T, e, 0, (n + 1) * mg) W.AT[e] = W.G[e];
    PAR_FOR(T, i, 0, n) W.bb[i] = 0.0;
    if (T.tid() == 0) W.bb[n] = 1.0;
    T.sync();
    double rnorm = 0.0;
    int mode = nnls(T, W, n + 1, mg, mg, &W.sc[2]);
    T.sync();
    rnorm = W.sc[2];
    if (mode != 1) return mode;
    if (rnorm <= 0.0) return 4;
    if (T.tid() == 0) {
        double t = 0.0;
        for (int i = 0; i < mg; ++i) t = t + W.G[i + (size_t)n * mg] * W.xn[i];       // ddot(m, h, y)
        const double fac = 1.0 - t;
        volatile double d = 1.0 + fac;
        W.isc[9] = (d - 1.0 > 0.0) ? 1 : 0;
        W.sc[3] = fac;
    }
    T.sync();
    if (!W.isc[9]) return 4;
    const double fac = 1.0 / W.sc[3];
    PAR_FOR(T, j, 0, n) {
        double t = 0.0;
        for (int i = 0; i < mg; ++i) t = t + W.G[i + (size_t)j * mg] * W.xn[i];
        W.xs[j] = fac * t;
    }
    PAR_FOR(T, i, 0, mg) y[i] = 0.0 + fac * W.xn[i];
    T.sync();
    // ---- back to the original variables: x = e^-1 (x + f), then the box (thread 0: a recurrence)
    if (T.tid() == 0) {
        for (int i = 0; i < n; ++i) W.xs[i] = W.xs[i] + 1.0 * W.ff[i];
        for (int i = n; i >= 1; --i) {
            double t = 0.0;
            for (int j = i + 1; j <= n; ++j) t = t + E(i, j) * W.xs[j - 1];
            W.xs[i - 1] = (W.xs[i - 1] - t) / E(i, i);
        }
        for (int i = 0; i < n; ++i) {
            double v = W.xs[i];
            if (v < lo[i]) v = lo[i];
            else if (v > hi[i]) v = hi[i];
            s[i] = v;
        }
    }
    T.sync();
    return 1;
#undef E
#undef G
}

// Rank-one update of the LDL' factorisation a (packed by columns) with z and sigma; w is work (Fletcher &
// Powell's composite-t method as in slsqp_core.c's ldl).  The outer recurrence runs in step on every thread
// (each recomputes the scalars from values that are final after the previous barrier); the inner loops
// over the rows below the diagonal are the parallel part.  New diagonals are parked in W.acc[n..2n) and
// written back at the end, so that no thread reads a diagonal another one is replacing.
template <class Team>
SPMD_HD void ldl(const Team &T, const Work &W, int n, double *a, double *z, double sigma, double *w)
{
    if (sigma == 0.0) return;
    double *dnew = W.acc + n;
    double t = 1.0 / sigma;
    int ij = 1;
    if (sigma <= 0.0) {
        PAR_FOR(T, i, 0, n) w[i] = z[i];
        T.sync();
        for (int i = 1; i <= n; ++i) {
            const double v = w[i - 1];
            t = t + v * v / a[ij - 1];
            PAR_FOR(T, j, i + 1, n + 1) w[j - 1] = w[j - 1] - v * a[ij - 1 + (j - i)];
            T.sync();
            ij = ij + (n - i) + 1;
        }
        if (t >= 0.0) t = 2.22e-16 / sigma;
        if (T.tid() == 0) {
            int q = ij;
            double tt = t;
            for (int i = 1; i <= n; ++i) {
                const int j = n + 1 - i;
                q = q - i;
                const double u = w[j - 1];
                w[j - 1] = tt;
                tt = tt - u * u / a[q - 1];
            }
            W.sc[4] = tt;
        }
        T.sync();
        t = W.sc[4];
        ij = ij - (n * (n + 1)) / 2;
        T.sync();
    }
    for (int i = 1; i <= n; ++i) {
        const double v = z[i - 1], aij = a[ij - 1];
        const double delta = v / aij;
        double tp = 0.0;
        if (sigma < 0.0) tp = w[i - 1];
        if (sigma > 0.0) tp = t + delta * v;
        const double alpha = tp / t;
        if (T.tid() == 0) dnew[i - 1] = alpha * aij;
        if (i == n) break;
        const double beta = delta / tp;
        if (alpha > 4.0) {
            const double gamma = t / tp;
            PAR_FOR(T, j, i + 1, n + 1) {
                const int p = ij - 1 + (j - i);
                const double u = a[p];
                a[p] = gamma * u + beta * z[j - 1];
                z[j - 1] = z[j - 1] - v * u;
            }
        } else {
            PAR_FOR(T, j, i + 1, n + 1) {
                const int p = ij - 1 + (j - i);
                z[j - 1] = z[j - 1] - v * a[p];
                a[p] = a[p] + beta * z[j - 1];
            }
        }
        T.sync();
        ij = ij + (n - i) + 1;
        t = tp;
    }
    T.sync();
    PAR_FOR(T, i, 0, n) a[i * n - (i * (i - 1)) / 2] = dnew[i];
    T.sync();
}

// The modified BFGS update of SLSQP's major iteration (the block after the derivatives arrive, before the
// next QP): u = g - (dc/dx)^T r - v, v = L D L' s, damping, then two rank-one updates of l.
// v holds the previous Lagrangian gradient on entry and is scratch afterwards; u [n] is scratch.
template <class Team>
SPMD_HD void bfgs_update(const Team &T, const Work &W, int H, double *l, const double *g, const double *nz, const double *r,
                         double *v, const double *s, double *u)
{
    const int n = 6 * H;
    double *vold = W.acc;
    PAR_FOR(T, i, 0, n) {
        const int j = i / 6, a = i % 6;
        const int tri = j * H - (j * (j - 1)) / 2 - j - 1;
        double t = 0.0;                                   // column i of dc/dx times r, rows in ascending order
        for (int kk = j + 1; kk <= H; ++kk) {
            t = t + nz[6 * H + 6 * (tri + kk) + a] * r[12 * kk + a];
            if (kk == j + 1) t = t + nz[i] * r[12 * (j + 1) + 6 + a];
        }
        u[i] = g[i] - t - v[i];
    }
    T.sync();
    PAR_FOR(T, i, 0, n) {
        const int off = i * n - (i * (i - 1)) / 2;
        double h1 = 0.0;
        for (int j = i + 1; j < n; ++j) h1 = h1 + l[off + (j - i)] * s[j];
        double vi = s[i] + h1;
        vi = l[off] * vi;
        vold[i] = vi;
    }
    T.sync();
    PAR_FOR(T, i, 0, n) {
        double h1 = 0.0;
        for (int j = 0; j < i; ++j) h1 = h1 + l[j * n - (j * (j - 1)) / 2 + (i - j)] * vold[j];
        v[i] = vold[i] + h1;
    }
    T.sync();
    if (T.tid() == 0) {
        double h1 = 0.0, h2 = 0.0;
        for (int i = 0; i < n; ++i) h1 = h1 + s[i] * u[i];
        for (int i = 0; i < n; ++i) h2 = h2 + s[i] * v[i];
        const double h3 = 0.2 * h2;
        double h4 = 0.0;
        int damp = 0;
        if (h1 < h3) {
            h4 = (h2 - h3) / (h2 - h1);
            h1 = h3;
            damp = 1;
        }
        W.sc[5] = h1; W.sc[6] = h2; W.sc[7] = h4; W.isc[10] = damp;
    }
    T.sync();
    const double h1 = W.sc[5], h2 = W.sc[6], h4 = W.sc[7];
    if (W.isc[10]) {
        const double da = 1.0 - h4;
        PAR_FOR(T, i, 0, n) {
            double ui = h4 * u[i];
            if (da != 0.0) ui = ui + da * v[i];
            u[i] = ui;
        }
        T.sync();
    }
    ldl(T, W, n, l, u, +1.0 / h1, v);
    ldl(T, W, n, l, v, -1.0 / h2, u);
}

}  // namespace lsqspmd
