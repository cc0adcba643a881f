/* Host-side SLSQP core of libmpc_slsqp.so.
 * Restatement in C of the SLSQP core (D. Kraft, "A software package for sequential quadratic
 * programming", DFVLR-FB 88-28, 1988; ACM TOMS 733) with the least-squares machinery of Lawson & Hanson,
 * "Solving Least Squares Problems" (1974): LSQ -> LSEI -> LSI -> LDP -> NNLS, H12, LDL.  Reverse
 * communication exactly like scipy.optimize._slsqp.slsqp (scipy <= 1.15): same arguments, same workspace
 * layout, so that a run can be compared call by call with scipy's compiled core.
 * 1-based index macros keep the text close to the published algorithms.
 * Pinned: tests/test_native_slsqp.py drives this file and scipy's core side by side on MPC problems and
 * requires identical iterates (compiled with -ffp-contract=off, the two agree bit for bit).
 * Not restated: equality constraints (meq > 0) and the exact line search (acc < 0); the MPC problem has
 * neither (ur_mpc.py:188-200: inequality constraints only, ftol > 0). */
#include <math.h>
#include <string.h>

#include "slsqp_core.h"

#define MAX(a, b) ((a) > (b) ? (a) : (b))
#define MIN(a, b) ((a) < (b) ? (a) : (b))

static const double epmach = 2.22e-16;

/* The hot loops are element-wise over contiguous runs; a second copy of these functions compiled for AVX2
 * is picked at load time where the CPU has it (no FMA contraction in either copy: -ffp-contract=off, so
 * both produce the same bits). */
#if defined(__x86_64__) && defined(__GNUC__) && !defined(MPC_SLSQP_NO_CLONES)
#define HOT __attribute__((target_clones("default", "avx2")))
#else
#define HOT
#endif

HOT static double ddot_sl(int n, const double *dx, int incx, const double *dy, int incy)
{
    double t = 0.0;
    if (n <= 0) return 0.0;
    if (incx == 1 && incy == 1) {
        int m = n % 5, i;
        for (i = 0; i < m; ++i) t = t + dx[i] * dy[i];
        for (i = m; i < n; i += 5)
            t = t + dx[i] * dy[i] + dx[i + 1] * dy[i + 1] + dx[i + 2] * dy[i + 2] + dx[i + 3] * dy[i + 3] + dx[i + 4] * dy[i + 4];
        return t;
    }
    {
        int ix = 0, iy = 0, i;
        if (incx < 0) ix = (-n + 1) * incx;
        if (incy < 0) iy = (-n + 1) * incy;
        for (i = 0; i < n; ++i) { t = t + dx[ix] * dy[iy]; ix += incx; iy += incy; }
    }
    return t;
}

HOT static void daxpy_sl(int n, double da, const double *dx, int incx, double *dy, int incy)
{
    int i, ix = 0, iy = 0;
    if (n <= 0 || da == 0.0) return;
    for (i = 0; i < n; ++i) { dy[iy] = dy[iy] + da * dx[ix]; ix += incx; iy += incy; }
}

static void dcopy_(int n, const double *dx, int incx, double *dy, int incy)
{
    int i, ix = 0, iy = 0;
    for (i = 0; i < n; ++i) { dy[iy] = dx[ix]; ix += incx; iy += incy; }
}

static void dscal_sl(int n, double da, double *dx, int incx)
{
    int i, ix = 0;
    for (i = 0; i < n; ++i) { dx[ix] = da * dx[ix]; ix += incx; }
}

static double dnrm2_(int n, const double *dx, int incx)
{
    /* scaled sum of squares */
    double scale = 0.0, ssq = 1.0;
    int i, ix = 0;
    if (n < 1) return 0.0;
    for (i = 0; i < n; ++i, ix += incx) {
        if (dx[ix] != 0.0) {
            double ax = fabs(dx[ix]);
            if (scale < ax) { ssq = 1.0 + ssq * (scale / ax) * (scale / ax); scale = ax; }
            else ssq = ssq + (ax / scale) * (ax / scale);
        }
    }
    return scale * sqrt(ssq);
}

static void dsrotg(double *da, double *db, double *c, double *s)
{
    double roe = *db, scale, r, z;
    if (fabs(*da) > fabs(*db)) roe = *da;
    scale = fabs(*da) + fabs(*db);
    if (scale == 0.0) { *c = 1.0; *s = 0.0; r = 0.0; }
    else {
        r = scale * sqrt((*da / scale) * (*da / scale) + (*db / scale) * (*db / scale));
        r = (roe >= 0.0 ? 1.0 : -1.0) * r;
        *c = *da / r;
        *s = *db / r;
    }
    z = *s;
    if (fabs(*c) > 0.0 && fabs(*c) <= *s) z = 1.0 / *c;
    *da = r;
    *db = z;
}

HOT static void dsrot(int n, double *dx, int incx, double *dy, int incy, double c, double s)
{
    int i, ix = 0, iy = 0;
    for (i = 0; i < n; ++i) {
        double t = c * dx[ix] + s * dy[iy];
        dy[iy] = c * dy[iy] - s * dx[ix];
        dx[ix] = t;
        ix += incx; iy += incy;
    }
}

/* Householder transformation (Lawson & Hanson H12).  u(1,j) = u[(j-1)*iue]; c addressed 1-based. */
HOT static void h12(int mode, int lpivot, int l1, int m, double *u, int iue, double *up, double *c, int ice, int icv, int ncv)
{
#define U(j) u[((j) - 1) * iue]
    double cl, clinv, sm, b;
    int i, j, i2, i3, i4, incr;
    if (0 >= lpivot || lpivot >= l1 || l1 > m) return;
    cl = fabs(U(lpivot));
    if (mode != 2) {
        for (j = l1; j <= m; ++j) { sm = fabs(U(j)); cl = MAX(sm, cl); }
        if (cl <= 0.0) return;
        clinv = 1.0 / cl;
        sm = (U(lpivot) * clinv) * (U(lpivot) * clinv);
        for (j = l1; j <= m; ++j) sm = sm + (U(j) * clinv) * (U(j) * clinv);
        cl = cl * sqrt(sm);
        if (U(lpivot) > 0.0) cl = -cl;
        *up = U(lpivot) - cl;
        U(lpivot) = cl;
    } else if (cl <= 0.0) return;
    if (ncv <= 0) return;
    b = *up * U(lpivot);
    if (b >= 0.0) return;
    b = 1.0 / b;
    i2 = 1 - icv + ice * (lpivot - 1);
    incr = ice * (l1 - lpivot);
    for (j = 1; j <= ncv; ++j) {
        i2 = i2 + icv;
        i3 = i2 + incr;
        i4 = i3;
        sm = c[i2 - 1] * *up;
        for (i = l1; i <= m; ++i) { sm = sm + c[i3 - 1] * U(i); i3 = i3 + ice; }
        if (sm == 0.0) continue;
        sm = sm * b;
        c[i2 - 1] = c[i2 - 1] + sm * *up;
        for (i = l1; i <= m; ++i) { c[i4 - 1] = c[i4 - 1] + sm * U(i); i4 = i4 + ice; }
    }
#undef U
}

/* Non-negative least squares (Lawson & Hanson NNLS): min ||a x - b||, x >= 0.  a(mda, n) column-major. */
HOT static void nnls(double *a, int mda, int m, int n, double *b, double *x, double *rnorm, double *w, double *z, int *index, int *mode)
{
#define A(i, j) a[((i) - 1) + ((j) - 1) * mda]
    const double factor = 1.0e-2;
    int i, ii, ip, iter, itmax, iz, iz1, iz2, izmax = 0, j, jj = 0, jz, k, l, npp1, nsetp;
    double alpha, asave, c, s, t, unorm, up, wmax;
    *mode = 2;
    if (m <= 0 || n <= 0) return;
    *mode = 1;
    iter = 0;
    itmax = 3 * n;
    for (i = 1; i <= n; ++i) index[i - 1] = i;
    iz1 = 1; iz2 = n; nsetp = 0; npp1 = 1;
    for (i = 0; i < n; ++i) x[i] = 0.0;
L110:
    if (iz1 > iz2 || nsetp >= m) goto L280;
    for (iz = iz1; iz <= iz2; ++iz) {
        j = index[iz - 1];
        w[j - 1] = ddot_sl(m - nsetp, &A(npp1, j), 1, &b[npp1 - 1], 1);
    }
L130:
    wmax = 0.0;
    for (iz = iz1; iz <= iz2; ++iz) {
        j = index[iz - 1];
        if (w[j - 1] <= wmax) continue;
        wmax = w[j - 1];
        izmax = iz;
    }
    if (wmax <= 0.0) goto L280;
    iz = izmax;
    j = index[iz - 1];
    asave = A(npp1, j);
    h12(1, npp1, npp1 + 1, m, &A(1, j), 1, &up, z, 1, 1, 0);
    unorm = dnrm2_(nsetp, &A(1, j), 1);
    t = factor * fabs(A(npp1, j));
    {
        volatile double d = unorm + t;
        if (d - unorm <= 0.0) goto L150;
    }
    dcopy_(m, b, 1, z, 1);
    h12(2, npp1, npp1 + 1, m, &A(1, j), 1, &up, z, 1, 1, 1);
    if (z[npp1 - 1] / A(npp1, j) > 0.0) goto L160;
L150:
    A(npp1, j) = asave;
    w[j - 1] = 0.0;
    goto L130;
L160:
    dcopy_(m, z, 1, b, 1);
    index[iz - 1] = index[iz1 - 1];
    index[iz1 - 1] = j;
    iz1 = iz1 + 1;
    nsetp = npp1;
    npp1 = npp1 + 1;
    for (jz = iz1; jz <= iz2; ++jz) {
        jj = index[jz - 1];
        h12(2, nsetp, npp1, m, &A(1, j), 1, &up, &A(1, jj), 1, mda, 1);
    }
    k = MIN(npp1, mda);
    w[j - 1] = 0.0;
    for (i = 0; i < m - nsetp; ++i) (&A(k, j))[i] = 0.0;
L180:
    for (ip = nsetp; ip >= 1; --ip) {
        if (ip != nsetp) daxpy_sl(ip, -z[ip], &A(1, jj), 1, z, 1);
        jj = index[ip - 1];
        z[ip - 1] = z[ip - 1] / A(ip, jj);
    }
    iter = iter + 1;
    if (iter > itmax) { *mode = 3; goto L280; }
    alpha = 1.0;
    jj = 0;
    for (ip = 1; ip <= nsetp; ++ip) {
        if (z[ip - 1] > 0.0) continue;
        l = index[ip - 1];
        t = -x[l - 1] / (z[ip - 1] - x[l - 1]);
        if (alpha < t) continue;
        alpha = t;
        jj = ip;
    }
    for (ip = 1; ip <= nsetp; ++ip) {
        l = index[ip - 1];
        x[l - 1] = (1.0 - alpha) * x[l - 1] + alpha * z[ip - 1];
    }
    if (jj == 0) goto L110;
    i = index[jj - 1];
L250:
    x[i - 1] = 0.0;
    jj = jj + 1;
    for (j = jj; j <= nsetp; ++j) {
        ii = index[j - 1];
        index[j - 2] = ii;
        dsrotg(&A(j - 1, ii), &A(j, ii), &c, &s);
        t = A(j - 1, ii);
        dsrot(n, &A(j - 1, 1), mda, &A(j, 1), mda, c, s);
        A(j - 1, ii) = t;
        A(j, ii) = 0.0;
        dsrot(1, &b[j - 2], 1, &b[j - 1], 1, c, s);
    }
    npp1 = nsetp;
    nsetp = nsetp - 1;
    iz1 = iz1 - 1;
    index[iz1 - 1] = i;
    if (nsetp <= 0) { *mode = 3; goto L280; }
    for (jj = 1; jj <= nsetp; ++jj) {
        i = index[jj - 1];
        if (x[i - 1] <= 0.0) goto L250;
    }
    dcopy_(m, b, 1, z, 1);
    goto L180;
L280:
    k = MIN(npp1, m);
    *rnorm = dnrm2_(m - nsetp, &b[k - 1], 1);
    if (npp1 > m) for (i = 0; i < n; ++i) w[i] = 0.0;
#undef A
}

/* Least distance programming: min ||x|| s.t. g x >= h.  g(mg, n) column-major with m rows used. */
HOT static void ldp(double *g, int mg, int m, int n, double *h, double *x, double *xnorm, double *w, int *index, int *mode)
{
#define G(i, j) g[((i) - 1) + ((j) - 1) * mg]
    int i, j, iw, iff, iz, iy, iwdual, n1;
    double fac, rnorm;
    *mode = 2;
    if (n <= 0) return;
    *mode = 1;
    for (i = 0; i < n; ++i) x[i] = 0.0;
    *xnorm = 0.0;
    if (m == 0) return;
    iw = 0;
    for (j = 1; j <= m; ++j) {
        for (i = 1; i <= n; ++i) { iw = iw + 1; w[iw - 1] = G(j, i); }
        iw = iw + 1;
        w[iw - 1] = h[j - 1];
    }
    iff = iw + 1;
    for (i = 1; i <= n; ++i) { iw = iw + 1; w[iw - 1] = 0.0; }
    w[iw] = 1.0;
    n1 = n + 1;
    iz = iw + 2;
    iy = iz + n1;
    iwdual = iy + m;
    nnls(w, n1, n1, m, &w[iff - 1], &w[iy - 1], &rnorm, &w[iwdual - 1], &w[iz - 1], index, mode);
    if (*mode != 1) return;
    *mode = 4;
    if (rnorm <= 0.0) return;
    fac = 1.0 - ddot_sl(m, h, 1, &w[iy - 1], 1);
    {
        volatile double d = 1.0 + fac;
        if (!(d - 1.0 > 0.0)) return;
    }
    *mode = 1;
    fac = 1.0 / fac;
    for (j = 1; j <= n; ++j) x[j - 1] = fac * ddot_sl(m, &G(1, j), 1, &w[iy - 1], 1);
    *xnorm = dnrm2_(n, x, 1);
    for (i = 0; i < m; ++i) w[i] = 0.0;
    daxpy_sl(m, fac, &w[iy - 1], 1, w, 1);
#undef G
}

/* Inequality-constrained least squares: min ||e x - f|| s.t. g x >= h. */
HOT static void lsi(double *e, double *f, double *g, double *h, int le, int me, int lg, int mg, int n, double *x, double *xnorm,
                double *w, int *jw, int *mode)
{
#define E(i, j) e[((i) - 1) + ((j) - 1) * le]
#define G(i, j) g[((i) - 1) + ((j) - 1) * lg]
    int i, j, k, triangular = 1, regular = 1;
    double t;
    /* The caller (lsq) hands over e = sqrt(d) l^T, which is upper triangular with exact zeros below the
     * diagonal.  Every term those zeros contribute to a Householder step is an exact zero, so leaving them
     * out changes no rounded value: the QR "factorisation" reduces to the pivot arithmetic, O(n^2) instead
     * of O(n^3).  Anything else takes the general route. */
    for (j = 1; j <= n && triangular; ++j)
        for (i = j + 1; i <= me; ++i)
            if (E(i, j) != 0.0) { triangular = 0; break; }
    if (triangular) {
        for (i = 1; i <= n; ++i) {
            double cl = fabs(E(i, i)), clinv, sm, b, up;
            if (i + 1 > me) continue;            /* h12: l1 > m, nothing to do */
            if (cl <= 0.0) continue;
            clinv = 1.0 / cl;
            sm = (E(i, i) * clinv) * (E(i, i) * clinv);
            cl = cl * sqrt(sm);
            if (E(i, i) > 0.0) cl = -cl;
            up = E(i, i) - cl;
            E(i, i) = cl;
            b = up * E(i, i);
            if (b >= 0.0) continue;
            b = 1.0 / b;
            for (j = i + 1; j <= n; ++j) {
                sm = E(i, j) * up;
                if (sm == 0.0) continue;
                sm = sm * b;
                E(i, j) = E(i, j) + sm * up;
            }
            sm = f[i - 1] * up;
            if (sm != 0.0) {
                sm = sm * b;
                f[i - 1] = f[i - 1] + sm * up;
            }
        }
    } else {
        for (i = 1; i <= n; ++i) {
            j = MIN(i + 1, n);
            h12(1, i, i + 1, me, &E(1, i), 1, &t, &E(1, j), 1, le, n - i);
            h12(2, i, i + 1, me, &E(1, i), 1, &t, f, 1, 1, 1);
        }
    }
    *mode = 5;
    for (j = 1; j <= n; ++j)
        if (!(fabs(E(j, j)) >= epmach)) regular = 0;
    if (regular) {
        /* g <- g r^-1 and h <- h - g f, column by column with all rows of g at once (a column of g is
         * contiguous): per entry the same products are added in the same order as in the row-by-row text
         * of Lawson & Hanson, so every rounded value is the same; only the memory walk differs. */
        double acc[mg > 0 ? mg : 1];
        /* Rows of g that start with exact zeros (the +-identity rows of the box bounds, the velocity rows)
         * keep them through the transformation, and a product with an exact zero adds nothing to a sum that
         * started at +0: blocks of rows whose first non-zero lies right of the current columns are skipped. */
        enum { RB = 16 };
        const int nblk = (mg + RB - 1) / RB;
        int first[nblk > 0 ? nblk : 1];
        for (k = 0; k < nblk; ++k) first[k] = n + 1;
        for (j = n; j >= 1; --j) {
            const double *gj = &G(1, j);
            for (i = 0; i < mg; ++i)
                if (gj[i] != 0.0) first[i / RB] = j;
        }
        for (j = 1; j <= n; ++j) {
            const double ejj = E(j, j);
            double *gj = &G(1, j);
            for (i = 0; i < mg; ++i) acc[i] = 0.0;
            for (k = 1; k + 3 < j; k += 4) {     /* four terms per pass over the rows, added in the same order */
                const double e0 = E(k, j), e1 = E(k + 1, j), e2 = E(k + 2, j), e3 = E(k + 3, j);
                const double *g0 = &G(1, k), *g1 = &G(1, k + 1), *g2 = &G(1, k + 2), *g3 = &G(1, k + 3);
                for (int blk = 0; blk < nblk; ++blk) {
                    if (first[blk] > k + 3) continue;
                    const int hi = MIN(mg, (blk + 1) * RB);
                    for (i = blk * RB; i < hi; ++i) {
                        double a = acc[i];
                        a = a + g0[i] * e0;
                        a = a + g1[i] * e1;
                        a = a + g2[i] * e2;
                        a = a + g3[i] * e3;
                        acc[i] = a;
                    }
                }
            }
            for (; k < j; ++k) {
                const double ekj = E(k, j);
                const double *gk = &G(1, k);
                for (i = 0; i < mg; ++i) acc[i] = acc[i] + gk[i] * ekj;
            }
            for (i = 0; i < mg; ++i) gj[i] = (gj[i] - acc[i]) / ejj;
        }
        for (i = 0; i < mg; ++i) acc[i] = 0.0;
        for (j = 1; j <= n; ++j) {
            const double fj = f[j - 1];
            const double *gj = &G(1, j);
            for (i = 0; i < mg; ++i) acc[i] = acc[i] + gj[i] * fj;
        }
        for (i = 0; i < mg; ++i) h[i] = h[i] - acc[i];
    } else {
        for (i = 1; i <= mg; ++i) {
            for (j = 1; j <= n; ++j) {
                if (!(fabs(E(j, j)) >= epmach)) return;
                G(i, j) = (G(i, j) - ddot_sl(j - 1, &G(i, 1), lg, &E(1, j), 1)) / E(j, j);
            }
            h[i - 1] = h[i - 1] - ddot_sl(n, &G(i, 1), lg, f, 1);
        }
    }
    ldp(g, lg, mg, n, h, x, xnorm, w, jw, mode);
    if (*mode != 1) return;
    daxpy_sl(n, 1.0, f, 1, x, 1);
    for (i = n; i >= 1; --i) {
        j = MIN(i + 1, n);
        x[i - 1] = (x[i - 1] - ddot_sl(n - i, &E(i, j), le, &x[j - 1], 1)) / E(i, i);
    }
    j = MIN(n + 1, me);
    t = dnrm2_(me - n, &f[j - 1], 1);
    *xnorm = sqrt(*xnorm * *xnorm + t * t);
#undef E
#undef G
}

/* QP subproblem in least-squares form (Kraft): e = sqrt(d) l^T, f = -d^(-1/2) l^(-1) g. */
HOT static void lsq(int m, int meq, int n, int nl, int la, double *l, double *g, double *a, double *b, double *xl, double *xu,
                double *x, double *y, double *w, int *jw, int *mode)
{
    int i, j, i1, i2, i3, i4, ie, iff, ic, id, ig, ih, il, ip, iw, m1, mineq, n1, n2, n3, nancnt;
    double diag, xnorm;
    n1 = n + 1;
    mineq = m - meq;
    m1 = mineq + n + n;
    n2 = n1 * n / 2 + 1;
    n2 = (n2 == nl) ? 0 : 1;
    n3 = n - n2;
    i2 = 1; i3 = 1; i4 = 1;
    ie = 1;
    iff = n * n + 1;
    for (i = 1; i <= n3; ++i) {
        i1 = n1 - i;
        diag = sqrt(l[i2 - 1]);
        for (j = 0; j < i1; ++j) w[i3 - 1 + j] = 0.0;
        dcopy_(i1 - n2, &l[i2 - 1], 1, &w[i3 - 1], n);
        dscal_sl(i1 - n2, diag, &w[i3 - 1], n);
        w[i3 - 1] = diag;
        w[iff - 2 + i] = (g[i - 1] - ddot_sl(i - 1, &w[i4 - 1], 1, &w[iff - 1], 1)) / diag;
        i2 = i2 + i1 - n2;
        i3 = i3 + n1;
        i4 = i4 + n;
    }
    if (n2 == 1) {
        w[i3 - 1] = l[nl - 1];
        for (j = 0; j < n3; ++j) w[i4 - 1 + j] = 0.0;
        w[iff - 2 + n] = 0.0;
    }
    dscal_sl(n, -1.0, &w[iff - 1], 1);
    ic = iff + n;
    id = ic + meq * n;
    ig = id + meq;
    for (i = 1; i <= mineq; ++i) dcopy_(n, &a[(meq + i - 1)], la, &w[ig - 2 + i], m1);
    ih = ig + m1 * n;
    iw = ih + mineq + 2 * n;
    if (mineq > 0) {
        dcopy_(mineq, &b[meq], 1, &w[ih - 1], 1);
        dscal_sl(mineq, -1.0, &w[ih - 1], 1);
    }
    ip = ig + mineq;
    il = ih + mineq;
    nancnt = 0;
    for (i = 1; i <= n; ++i) {
        if (xl[i - 1] == xl[i - 1]) {
            w[il - 1] = xl[i - 1];
            for (j = 1; j <= n; ++j) w[ip - 1 + m1 * (j - 1)] = 0.0;
            w[ip - 1 + m1 * (i - 1)] = 1.0;
            ip = ip + 1; il = il + 1;
        } else nancnt = nancnt + 1;
    }
    for (i = 1; i <= n; ++i) {
        if (xu[i - 1] == xu[i - 1]) {
            w[il - 1] = -xu[i - 1];
            for (j = 1; j <= n; ++j) w[ip - 1 + m1 * (j - 1)] = 0.0;
            w[ip - 1 + m1 * (i - 1)] = -1.0;
            ip = ip + 1; il = il + 1;
        } else nancnt = nancnt + 1;
    }
    (void)ic; (void)id; (void)ie;
    if (meq != 0) { *mode = 2; return; }   /* equality constraints are not part of this restatement */
    /* No equality constraints: LSEI would only copy e, f and g once more and call LSI on the copies (they are
     * not read again here), so LSI works on them in place; w(iw...) is LDP's workspace and returns the
     * multipliers, as in LSEI. */
    for (i = 0; i < m1 - nancnt; ++i) w[iw - 1 + i] = 0.0;
    lsi(&w[0], &w[iff - 1], &w[ig - 1], &w[ih - 1], n, n, m1, m1 - nancnt, n, x, &xnorm, &w[iw - 1], jw, mode);
    if (*mode == 1) {
        dcopy_(m, &w[iw - 1], 1, y, 1);
        if (n3 > 0) {
            double nan_ = NAN;
            for (i = m + 1; i <= m + n3 + n3; ++i) y[i - 1] = nan_;
        }
    }
    for (i = 0; i < n; ++i) {   /* bound(): keep the step inside the box */
        if (xl[i] == xl[i] && x[i] < xl[i]) x[i] = xl[i];
        else if (xu[i] == xu[i] && x[i] > xu[i]) x[i] = xu[i];
    }
}

static mpc_lsq_fn lsq_hook = 0;

void mpc_slsqp_lsq_builtin(int m, int meq, int n, int nl, int la, double *l, double *g, double *a, double *b, double *xl,
                           double *xu, double *x, double *y, double *w, int *jw, int *mode)
{
    lsq(m, meq, n, nl, la, l, g, a, b, xl, xu, x, y, w, jw, mode);
}

void mpc_slsqp_set_lsq(mpc_lsq_fn fn) { lsq_hook = fn; }

static void lsq_dispatch(int m, int meq, int n, int nl, int la, double *l, double *g, double *a, double *b, double *xl,
                         double *xu, double *x, double *y, double *w, int *jw, int *mode)
{
    if (lsq_hook) lsq_hook(m, meq, n, nl, la, l, g, a, b, xl, xu, x, y, w, jw, mode);
    else lsq(m, meq, n, nl, la, l, g, a, b, xl, xu, x, y, w, jw, mode);
}

/* Rank-one update of an LDL^T factorisation (Fletcher & Powell). */
static void ldl(int n, double *a, double *z, double sigma, double *w)
{
    int i, ij, j;
    double alpha, beta, delta, gamma, t, tp = 0.0, u, v;
    if (sigma == 0.0) return;
    ij = 1;
    t = 1.0 / sigma;
    if (sigma <= 0.0) {
        for (i = 1; i <= n; ++i) w[i - 1] = z[i - 1];
        for (i = 1; i <= n; ++i) {
            v = w[i - 1];
            t = t + v * v / a[ij - 1];
            for (j = i + 1; j <= n; ++j) { ij = ij + 1; w[j - 1] = w[j - 1] - v * a[ij - 1]; }
            ij = ij + 1;
        }
        if (t >= 0.0) t = epmach / sigma;
        for (i = 1; i <= n; ++i) {
            j = n + 1 - i;
            ij = ij - i;
            u = w[j - 1];
            w[j - 1] = t;
            t = t - u * u / a[ij - 1];
        }
    }
    for (i = 1; i <= n; ++i) {
        v = z[i - 1];
        delta = v / a[ij - 1];
        if (sigma < 0.0) tp = w[i - 1];
        if (sigma > 0.0) tp = t + delta * v;
        alpha = tp / t;
        a[ij - 1] = alpha * a[ij - 1];
        if (i == n) return;
        beta = delta / tp;
        if (alpha > 4.0) {
            gamma = t / tp;
            for (j = i + 1; j <= n; ++j) {
                ij = ij + 1;
                u = a[ij - 1];
                a[ij - 1] = gamma * u + beta * z[j - 1];
                z[j - 1] = z[j - 1] - v * u;
            }
        } else {
            for (j = i + 1; j <= n; ++j) {
                ij = ij + 1;
                z[j - 1] = z[j - 1] - v * a[ij - 1];
                a[ij - 1] = a[ij - 1] + beta * z[j - 1];
            }
        }
        ij = ij + 1;
        t = tp;
    }
}


static mpc_bfgs_fn bfgs_hook = 0;

void mpc_slsqp_set_bfgs(mpc_bfgs_fn fn) { bfgs_hook = fn; }

/* Modified BFGS update of the LDL' factor l (Powell's damping), the block of the body that follows a
 * derivative evaluation. */
void mpc_slsqp_bfgs_builtin(int m, int n, int la, double *l, double *g, double *a, double *r, double *u, double *v, double *s)
{
#define A(i, j) a[((i) - 1) + ((j) - 1) * la]
    const int n1 = n + 1;
    int i, j, k;
    double h1, h2, h3, h4;
    for (i = 1; i <= n; ++i) u[i - 1] = g[i - 1] - ddot_sl(m, &A(1, i), 1, r, 1) - v[i - 1];
    k = 0;
    for (i = 1; i <= n; ++i) {
        h1 = 0.0;
        k = k + 1;
        for (j = i + 1; j <= n; ++j) { k = k + 1; h1 = h1 + l[k - 1] * s[j - 1]; }
        v[i - 1] = s[i - 1] + h1;
    }
    k = 1;
    for (i = 1; i <= n; ++i) { v[i - 1] = l[k - 1] * v[i - 1]; k = k + n1 - i; }
    for (i = n; i >= 1; --i) {
        h1 = 0.0;
        k = i;
        for (j = 1; j <= i - 1; ++j) { h1 = h1 + l[k - 1] * v[j - 1]; k = k + n - j; }
        v[i - 1] = v[i - 1] + h1;
    }
    h1 = ddot_sl(n, s, 1, u, 1);
    h2 = ddot_sl(n, s, 1, v, 1);
    h3 = 0.2 * h2;
    if (h1 < h3) {
        h4 = (h2 - h3) / (h2 - h1);
        h1 = h3;
        dscal_sl(n, h4, u, 1);
        daxpy_sl(n, 1.0 - h4, v, 1, u, 1);
    }
    ldl(n, l, u, +1.0 / h1, v);
    ldl(n, l, v, -1.0 / h2, u);
#undef A
}

/* The loops of Kraft's body over the constraints (slsqp_body.inc): every chain of additions in the order of his
 * text.  The device has its own forms of the same chains (kernels_slsqp.hip). */
static void sl_fill(int n, double *x, double v)
{
    int i;
    for (i = 0; i < n; ++i) x[i] = v;
}
/* t + sum_j max(-c(j), c(j) for j <= meq, else 0) */
static double sl_sum_infeas(int m, int meq, const double *c, double t)
{
    int j;
    for (j = 1; j <= m; ++j) {
        const double h = (j <= meq) ? c[j - 1] : 0.0;
        t = t + MAX(-c[j - 1], h);
    }
    return t;
}
/* t + sum_j mu(j) max(-c(j), c(j) for j <= meq, else 0) */
static double sl_sum_mu_infeas(int m, int meq, const double *mu, const double *c, double t)
{
    int j;
    for (j = 1; j <= m; ++j) {
        const double h = (j <= meq) ? c[j - 1] : 0.0;
        t = t + mu[j - 1] * MAX(-c[j - 1], h);
    }
    return t;
}
/* t + sum_j |r(j)| |c(j)| */
static double sl_sum_abs_prod(int m, const double *r, const double *c, double t)
{
    int j;
    for (j = 1; j <= m; ++j) t = t + fabs(r[j - 1]) * fabs(c[j - 1]);
    return t;
}
/* mu(j) = max(|r(j)|, (mu(j) + |r(j)|) / 2) */
static void sl_mu_update(int m, const double *r, double *mu)
{
    int j;
    for (j = 1; j <= m; ++j) {
        const double h = fabs(r[j - 1]);
        mu[j - 1] = MAX(h, (mu[j - 1] + h) / 2.0);
    }
}

void mpc_slsqpb(int m, int meq, int la, int n, double *x, double *xl, double *xu, double f, double *c, double *g, double *a,
                   double *acc, int *iter, int *mode, double *r, double *l, double *x0, double *mu, double *s, double *u,
                   double *v, double *w, int *iw, slsqp_state *S)
#define SLSQP_BODY_FULL 1
#include "slsqp_body.inc"
#undef SLSQP_BODY_FULL

/* One reverse-communication call; arguments as scipy.optimize._slsqp.slsqp.  a is column-major (la, n+1). */
void mpc_slsqp_step(int m, int meq, int la, int n, double *x, double *xl, double *xu, double f, double *c, double *g, double *a,
                double *acc, int *iter, int *mode, double *w, int l_w, int *jw, int l_jw, slsqp_state *S)
{
    int n1 = n + 1, mineq = m - meq + n1 + n1, il, im, ix, ir, is, iu, iv, iw;
    il = (3 * n1 + m) * (n1 + 1) + (n1 - meq + 1) * (mineq + 2) + 2 * mineq + (n1 + mineq) * (n1 - meq) + 2 * meq + n1 * n / 2 +
         2 * m + 3 * n + 4 * n1 + 1;
    im = MAX(mineq, n1 - meq);
    if (l_w < il || l_jw < im) { *mode = 1000 * MAX(10, il) + MAX(10, im); return; }
    im = 1;
    il = im + la;
    ix = il + (n1 * n) / 2 + 1;
    ir = ix + n;
    is = ir + n + n + la;
    iu = is + n1;
    iv = iu + n1;
    iw = iv + n1;
    mpc_slsqpb(m, meq, la, n, x, xl, xu, f, c, g, a, acc, iter, mode, &w[ir - 1], &w[il - 1], &w[ix - 1], &w[im - 1], &w[is - 1],
           &w[iu - 1], &w[iv - 1], &w[iw - 1], jw, S);
}
