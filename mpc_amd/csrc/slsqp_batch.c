/* Batched host driver around the SLSQP core: see include/mpc_slsqp.h. */
#include <omp.h>
#include <stdlib.h>
#include <string.h>

#include "mpc_slsqp.h"
#include "slsqp_core.h"

struct mpc_slsqp_batch {
    int H, n, m, n1, nzc, cap, threads, count, ljw;
    size_t lw, ldl_len, r_len;
    /* per problem */
    double *x, *f, *c, *g, *nz, *mu, *l, *x0, *r, *s, *u, *v, *acc;
    int32_t *iter, *mode, *qpmode, *qpflags;
    int offload;
    unsigned char *done;
    slsqp_state *st;
    /* shared */
    double *lb, *ub;
    int32_t *scatter;         /* position of structured entry k inside the dense column-major a(m, n+1) */
    /* per thread */
    double *wscr, *ascr, *tsec;
    int *jwscr;
    double wall;
};

int mpc_slsqp_abi_version(void) { return MPC_SLSQP_ABI_VERSION; }

static void *zalloc(size_t count, size_t size) { return calloc(count ? count : 1, size); }

void mpc_slsqp_destroy(mpc_slsqp_batch *b)
{
    if (!b) return;
    free(b->x); free(b->f); free(b->c); free(b->g); free(b->nz); free(b->mu); free(b->l); free(b->x0); free(b->r);
    free(b->s); free(b->u); free(b->v); free(b->acc); free(b->iter); free(b->mode); free(b->qpmode); free(b->qpflags); free(b->done); free(b->st);
    free(b->lb); free(b->ub); free(b->scatter); free(b->wscr); free(b->ascr); free(b->tsec); free(b->jwscr);
    free(b);
}

int mpc_slsqp_create(int horizon, int capacity, int threads, mpc_slsqp_batch **out)
{
    if (!out) return -1;
    *out = NULL;
    if (horizon <= 0 || horizon > 255 || capacity <= 0) return -1;
    mpc_slsqp_batch *b = (mpc_slsqp_batch *)calloc(1, sizeof(*b));
    if (!b) return -2;
    const int H = horizon, n = 6 * H, m = 12 * (H + 1), n1 = n + 1, meq = 0;
    const int mineq = m - meq + n1 + n1;
    b->H = H; b->n = n; b->m = m; b->n1 = n1; b->cap = capacity;
    b->nzc = 6 * H + 3 * H * (H + 1);
    b->threads = threads > 0 ? threads : omp_get_max_threads();
    /* workspace length of the Fortran driver (_slsqp_py.py:376-382); the persistent vectors live apart */
    b->lw = (size_t)(3 * n1 + m) * (n1 + 1) + (size_t)(n1 - meq + 1) * (mineq + 2) + 2 * (size_t)mineq +
            (size_t)(n1 + mineq) * (n1 - meq) + 2 * meq + (size_t)n1 * n / 2 + 2 * (size_t)m + 3 * (size_t)n + 4 * (size_t)n1 + 1;
    b->ljw = mineq > n1 - meq ? mineq : n1 - meq;
    b->ldl_len = (size_t)n1 * n / 2 + 1;
    b->r_len = (size_t)m + 2 * (size_t)n + 2;
    const size_t N = (size_t)capacity, T = (size_t)b->threads;
    b->x = zalloc(N * n, 8); b->f = zalloc(N, 8); b->c = zalloc(N * m, 8); b->g = zalloc(N * n1, 8);
    b->nz = zalloc(N * b->nzc, 8); b->mu = zalloc(N * m, 8); b->l = zalloc(N * b->ldl_len, 8); b->x0 = zalloc(N * n, 8);
    b->r = zalloc(N * b->r_len, 8); b->s = zalloc(N * n1, 8); b->u = zalloc(N * n1, 8); b->v = zalloc(N * n1, 8);
    b->acc = zalloc(N, 8); b->iter = zalloc(N, 4); b->mode = zalloc(N, 4); b->done = zalloc(N, 1);
    b->qpmode = zalloc(N, 4); b->qpflags = zalloc(N, 4);
    b->st = zalloc(N, sizeof(slsqp_state));
    b->lb = zalloc(n, 8); b->ub = zalloc(n, 8); b->scatter = zalloc(b->nzc, 4);
    b->wscr = zalloc(T * b->lw, 8); b->ascr = zalloc(T * (size_t)n1 * m, 8); b->tsec = zalloc(T, 8);
    b->jwscr = zalloc(T * (size_t)b->ljw, sizeof(int));
    if (!b->x || !b->f || !b->c || !b->g || !b->nz || !b->mu || !b->l || !b->x0 || !b->r || !b->s || !b->u || !b->v ||
        !b->acc || !b->iter || !b->mode || !b->qpmode || !b->qpflags || !b->done || !b->st || !b->lb || !b->ub || !b->scatter || !b->wscr || !b->ascr ||
        !b->tsec || !b->jwscr) {
        mpc_slsqp_destroy(b);
        return -2;
    }
    /* structured dc/dx -> dense a (include/mpc_amd.h, mpc_eval_fd_nz): a(row, col) at col * m + row */
    for (int j = 0; j < H; ++j) {
        const int tri = j * H - (j * (j - 1)) / 2 - j - 1;
        for (int a = 0; a < 6; ++a) {
            const int col = 6 * j + a;
            b->scatter[col] = col * m + 12 * (j + 1) + 6 + a;
            for (int k = j + 1; k <= H; ++k) b->scatter[6 * H + 6 * (tri + k) + a] = col * m + 12 * k + a;
        }
    }
    *out = b;
    return 0;
}

int mpc_slsqp_begin(mpc_slsqp_batch *b, int count, const double *x0, const double *lb, const double *ub, double acc,
                    int maxiter)
{
    if (!b || count < 0 || count > b->cap || (count > 0 && !x0) || !lb || !ub || maxiter < 1) return -1;
    const int n = b->n;
    memcpy(b->lb, lb, sizeof(double) * n);
    memcpy(b->ub, ub, sizeof(double) * n);
    b->count = count;
    for (int p = 0; p < count; ++p) {
        double *x = b->x + (size_t)p * n;
        for (int i = 0; i < n; ++i) {
            double v = x0[(size_t)p * n + i];
            v = v < lb[i] ? lb[i] : v;
            v = v > ub[i] ? ub[i] : v;
            x[i] = v;
        }
        b->acc[p] = acc;
        b->iter[p] = maxiter - 1;      /* majiter on entry = the iteration limit, maxiter - 1 (_slsqp_py.py:245,392) */
        b->mode[p] = 0;
        b->done[p] = 0;
        memset(&b->st[p], 0, sizeof(slsqp_state));
        b->st[p].offload = b->offload;
        b->qpmode[p] = 0;
        b->qpflags[p] = 0;
    }
    for (int t = 0; t < b->threads; ++t) b->tsec[t] = 0.0;
    b->wall = 0.0;
    return 0;
}

double *mpc_slsqp_x(mpc_slsqp_batch *b) { return b->x; }
double *mpc_slsqp_f(mpc_slsqp_batch *b) { return b->f; }
double *mpc_slsqp_c(mpc_slsqp_batch *b) { return b->c; }
double *mpc_slsqp_g(mpc_slsqp_batch *b) { return b->g; }
double *mpc_slsqp_nz(mpc_slsqp_batch *b) { return b->nz; }
const int32_t *mpc_slsqp_mode(mpc_slsqp_batch *b) { return b->mode; }
const int32_t *mpc_slsqp_iterations(mpc_slsqp_batch *b) { return b->iter; }

int mpc_slsqp_set_offload(mpc_slsqp_batch *b, int on)
{
    if (!b) return -1;
    b->offload = on < 0 ? 0 : (on > 2 ? 2 : on);      /* 2: the caller solves the augmented problem as well (slsqp_core.h) */
    return 0;
}
double *mpc_slsqp_lo(mpc_slsqp_batch *b) { return b->u; }
double *mpc_slsqp_hi(mpc_slsqp_batch *b) { return b->v; }
double *mpc_slsqp_s(mpc_slsqp_batch *b) { return b->s; }
double *mpc_slsqp_r(mpc_slsqp_batch *b) { return b->r; }
int32_t *mpc_slsqp_qp_mode(mpc_slsqp_batch *b) { return b->qpmode; }
const int32_t *mpc_slsqp_qp_flags(mpc_slsqp_batch *b) { return b->qpflags; }

double mpc_slsqp_core_seconds(const mpc_slsqp_batch *b)
{
    double mx = 0.0;
    for (int t = 0; t < b->threads; ++t) mx = b->tsec[t] > mx ? b->tsec[t] : mx;
    return mx;
}

int mpc_slsqp_advance_qp(mpc_slsqp_batch *b, int32_t *want_f, int32_t *n_want_f, int32_t *want_g, int32_t *n_want_g,
                         int32_t *want_q, int32_t *n_want_q)
{
    if (!b || !want_f || !n_want_f || !want_g || !n_want_g) return -1;
    if (b->offload && (!want_q || !n_want_q)) return -1;
    const int n = b->n, m = b->m, n1 = b->n1, nzc = b->nzc;
    const double t_round = omp_get_wtime();
#pragma omp parallel for schedule(dynamic, 1) num_threads(b->threads)
    for (int p = 0; p < b->count; ++p) {
        if (b->done[p]) continue;
        const int tid = omp_get_thread_num();
        const double t0 = omp_get_wtime();
        double *a = b->ascr + (size_t)tid * n1 * m;
        double *g = b->g + (size_t)p * n1;
        if (b->offload) {
            if (b->mode[p] == MPC_SLSQP_MODE_QP) {
                b->st[p].qp_mode = b->qpmode[p];
                b->st[p].qp_delta = b->r[(size_t)p * b->r_len + m];      /* an augmented problem's extra entry */
            }
            if (b->mode[p] != 1) g[n] = 0.0;
        } else if (b->mode[p] != 1) {
            /* fresh derivatives: the dense a(m, n+1) of this call, zero except the structured entries.  A call
             * that follows a line-search evaluation (mode 1) does not read a. */
            memset(a, 0, sizeof(double) * (size_t)n1 * m);
            const double *nz = b->nz + (size_t)p * nzc;
            for (int k = 0; k < nzc; ++k) a[b->scatter[k]] = nz[k];
            g[n] = 0.0;
        }
        int mode = b->mode[p], iter = b->iter[p];
        mpc_slsqpb(m, 0, m, n, b->x + (size_t)p * n, b->lb, b->ub, b->f[p], b->c + (size_t)p * m, g, a, &b->acc[p], &iter, &mode,
                   b->r + (size_t)p * b->r_len, b->l + (size_t)p * b->ldl_len, b->x0 + (size_t)p * n, b->mu + (size_t)p * m,
                   b->s + (size_t)p * n1, b->u + (size_t)p * n1, b->v + (size_t)p * n1, b->wscr + (size_t)tid * b->lw,
                   b->jwscr + (size_t)tid * b->ljw, &b->st[p]);
        b->mode[p] = mode;
        b->iter[p] = iter;
        b->qpflags[p] = MPC_QP_FLAGS_OF(&b->st[p]);
        if (mode != 1 && mode != -1 && mode != MPC_SLSQP_MODE_QP) b->done[p] = 1;
        b->tsec[tid] += omp_get_wtime() - t0;
    }
    int nf = 0, ng = 0, nq = 0;
    for (int p = 0; p < b->count; ++p) {
        if (b->done[p]) continue;
        if (b->mode[p] == 1) want_f[nf++] = p;
        else if (b->mode[p] == -1) want_g[ng++] = p;
        else if (want_q) want_q[nq++] = p;
    }
    *n_want_f = nf;
    *n_want_g = ng;
    if (n_want_q) *n_want_q = nq;
    b->wall += omp_get_wtime() - t_round;
    return nf + ng + nq;
}

int mpc_slsqp_pack(mpc_slsqp_batch *b, int nf, const int32_t *want_f, int ng, const int32_t *want_g, int nq,
                   const int32_t *want_q, double *stage, int32_t *istage)
{
    if (!b || !b->offload || nf < 0 || ng < 0 || nq < 0 || !stage || !istage) return -1;
    if ((nf && !want_f) || (ng && !want_g) || (nq && !want_q)) return -1;
    const int n = b->n, n1 = b->n1;
    const double *lb = b->lb, *ub = b->ub;
    double *xf = stage, *xg = xf + (size_t)nf * n, *lo = xg + (size_t)ng * n, *hi = lo + (size_t)nq * n, *sv = hi + (size_t)nq * n;
    int32_t *jf = istage, *jg = jf + nf, *jq = jg + ng, *fl = jq + nq;
    const int rows = nf + ng + nq;
#pragma omp parallel for schedule(static) num_threads(b->threads)
    for (int k = 0; k < rows; ++k) {
        if (k < nf + ng) {                                  /* the point to evaluate, inside the box (_clip_x_for_func) */
            const int p = k < nf ? want_f[k] : want_g[k - nf];
            const double *x = b->x + (size_t)p * n;
            double *dst = k < nf ? xf + (size_t)k * n : xg + (size_t)(k - nf) * n;
            for (int i = 0; i < n; ++i) {
                double v = x[i];
                v = v < lb[i] ? lb[i] : v;
                v = v > ub[i] ? ub[i] : v;
                dst[i] = v;
            }
            if (k < nf) jf[k] = p;
            else jg[k - nf] = p;
        } else {
            const int q = k - nf - ng, p = want_q[q];
            memcpy(lo + (size_t)q * n, b->u + (size_t)p * n1, sizeof(double) * n);
            memcpy(hi + (size_t)q * n, b->v + (size_t)p * n1, sizeof(double) * n);
            memcpy(sv + (size_t)q * n, b->s + (size_t)p * n1, sizeof(double) * n);
            jq[q] = p;
            fl[q] = b->qpflags[p];
        }
    }
    return 0;
}

int mpc_slsqp_unpack(mpc_slsqp_batch *b, int nf, const int32_t *want_f, int ng, const int32_t *want_g, int nq,
                     const int32_t *want_q, const double *stage, const int32_t *istage)
{
    if (!b || !b->offload || nf < 0 || ng < 0 || nq < 0 || !stage || !istage) return -1;
    if ((nf && !want_f) || (ng && !want_g) || (nq && !want_q)) return -1;
    const int n = b->n, m = b->m, n1 = b->n1;
    const double *cv = stage, *gv = cv + (size_t)nf * m, *sv = gv + (size_t)ng * n, *rv = sv + (size_t)nq * n, *fv = rv + (size_t)nq * m;
    const int rows = nf + ng + nq;
#pragma omp parallel for schedule(static) num_threads(b->threads)
    for (int k = 0; k < rows; ++k) {
        if (k < nf) {
            const int p = want_f[k];
            b->f[p] = fv[k];
            memcpy(b->c + (size_t)p * m, cv + (size_t)k * m, sizeof(double) * m);
        } else if (k < nf + ng) {
            const int q = k - nf, p = want_g[q];
            memcpy(b->g + (size_t)p * n1, gv + (size_t)q * n, sizeof(double) * n);
        } else {
            const int q = k - nf - ng, p = want_q[q];
            b->qpmode[p] = istage[q];
            if (istage[q] == 1) {
                memcpy(b->s + (size_t)p * n1, sv + (size_t)q * n, sizeof(double) * n);
                memcpy(b->r + (size_t)p * b->r_len, rv + (size_t)q * m, sizeof(double) * m);
            }
        }
    }
    return 0;
}

int mpc_slsqp_advance(mpc_slsqp_batch *b, int32_t *want_f, int32_t *n_want_f, int32_t *want_g, int32_t *n_want_g)
{
    if (!b || b->offload) return -1;
    return mpc_slsqp_advance_qp(b, want_f, n_want_f, want_g, n_want_g, 0, 0);
}
