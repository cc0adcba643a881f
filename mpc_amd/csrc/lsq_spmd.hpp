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
#include <hip/hip_runtime.h>
#include <type_traits>
#define SPMD_HD __host__ __device__
#else
#define SPMD_HD
#endif

#if defined(__HIPCC__)
#define SPMD_UNROLL _Pragma("unroll")
#else
#define SPMD_UNROLL
#endif

// Development only (-DMPC_QP_PROFILE, never in the shipped library): thread 0 of every team adds the shader cycles
// between two marks to a per-phase counter.
#if defined(MPC_QP_PROFILE) && defined(__HIPCC__)
__device__ unsigned long long mpc_qp_prof[32];
#endif
#if defined(MPC_QP_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
#define QP_MARK(T, id)                                                                       \
    do {                                                                                     \
        (T).sync();                                                                          \
        if ((T).tid() == 0) {                                                                \
            const unsigned long long now_ = __builtin_readcyclecounter();                    \
            atomicAdd(&mpc_qp_prof[id], now_ - *reinterpret_cast<unsigned long long *>(&W.sc[15])); \
            *reinterpret_cast<unsigned long long *>(&W.sc[15]) = now_;                       \
        }                                                                                    \
    } while (0)
#define QP_MARK_START(T)                                                                     \
    do {                                                                                     \
        if ((T).tid() == 0) *reinterpret_cast<unsigned long long *>(&W.sc[15]) = __builtin_readcyclecounter(); \
    } while (0)
#else
#define QP_MARK(T, id) do {} while (0)
#define QP_MARK_START(T) do {} while (0)
#endif

#include "lsq_wave.hpp"

namespace lsqspmd {

struct SerialTeam {
    static constexpr bool kWave = false;      // not a team of whole wavefronts: lsq_wave.hpp is never used
    static constexpr bool kLdsPool = false;
    SPMD_HD int tid() const { return 0; }
    SPMD_HD int size() const { return 1; }
    SPMD_HD int turn() const { return 0; }
    SPMD_HD int chain_lane() const { return 0; }
    SPMD_HD void sync() const {}
    SPMD_HD void wave_sync() const {}
};

#define PAR_FOR(T, i, lo, hi) for (int i = (lo) + (T).tid(); i < (hi); i += (T).size())

// Workspace of one team (doubles unless noted).  mg = m + 2n rows of [g h] and n1 = n + 1 columns (the variables
// and h) for the plain problem; Kraft's augmented problem (aug = 1: one variable more, two more bound rows) has
// mg + 2 and n + 2, and every size below is taken with those.  Two pools: the BIG one holds the two
// matrices and lives in global memory; the SMALL one (the packed triangle and every vector) is what the serial
// sections and the broadcast reads touch, and goes to LDS when it fits (kernels_qp.hip).
// (The plain problem keeps its own, smaller layout: carved with the augmented sizes, whole solves were 1 % slower.)
struct Work {
    double *G;     // [n1*mg]: G[i + j*mg] = g(i, j) for j < n; G[i + n*mg] = h(i)
    double *AT;    // [n1*mg]: the copy NNLS works on and overwrites (LDP needs g and h again afterwards)
    double *E;     // [n(n+1)/2 + 1]: on entry the packed LDL' factor l; lsq turns it IN PLACE into e = sqrt(d) l^T,
                   // e(r, c) for r <= c at E[off(r) + c - r], off(r) = r n - r(r-1)/2 (0-based), i.e. row by row;
                   // the last entry is the augmented problem's e(n, n) = rho (Kraft's l(n3)), unused otherwise
    double *ff;    // [n]
    double *acc;   // [max(mg, team size)]; NNLS's reduction (redv) borrows it
    double *bb;    // [n1]  NNLS right-hand side
    double *zz;    // [n1]
    double *xn;    // [mg]  NNLS solution (dual variables of LDP)
    double *wd;    // [mg]  NNLS dual vector
    double *piv;   // [n1]  copy of the pivot column
    double *xs;    // [n + aug]
    double *sc;    // [16]  scalars broadcast from thread 0
    double *redv;  // = acc
    int *redi;     // [max(team size, n1)]: NNLS's reduction; afterwards the final set, sorted
    int *idx;      // [mg]
    int *isc;      // [16]
};

SPMD_HD inline size_t big_doubles(int n, int m, int aug = 0)
{
    return 2 * ((size_t)n + 1 + aug) * ((size_t)m + 2 * (size_t)n + 2 * aug);
}
SPMD_HD inline size_t small_doubles(int n, int m, int team, int aug = 0)
{
    const size_t mg = (size_t)m + 2 * (size_t)n + 2 * aug, n1 = (size_t)n + 1 + aug;
    const size_t accn = mg > (size_t)team ? mg : (size_t)team;
    return (size_t)n * (n + 1) / 2 + 1 + n + accn + n1 + n1 + mg + mg + n1 + (n + aug) + 16;
}
SPMD_HD inline size_t work_doubles(int n, int m, int team, int aug = 0)
{
    return big_doubles(n, m, aug) + small_doubles(n, m, team, aug);
}
SPMD_HD inline size_t work_ints(int n, int m, int team, int aug = 0)
{
    return (size_t)(team > n + 1 + aug ? team : n + 1 + aug) + (size_t)m + 2 * (size_t)n + 2 * aug + 16;
}

SPMD_HD inline Work carve(double *big, double *d, int *i, int n, int m, int team, int aug = 0)
{
    const size_t mg = (size_t)m + 2 * (size_t)n + 2 * aug, n1 = (size_t)n + 1 + aug;
    Work W;
    W.G = big;
    W.AT = big + n1 * mg;
    W.E = d;            d += (size_t)n * (n + 1) / 2 + 1;
    W.ff = d;           d += n;
    W.acc = d;          d += mg > (size_t)team ? mg : (size_t)team;
    W.bb = d;           d += n1;
    W.zz = d;           d += n1;
    W.xn = d;           d += mg;
    W.wd = d;           d += mg;
    W.piv = d;          d += n1;
    W.xs = d;           d += n + aug;
    W.sc = d;
    W.redv = W.acc;
    W.redi = i;         i += (size_t)team > n1 ? (size_t)team : n1;
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

// t + p[0] + p[1] + ... in that order; the loads of a block are issued together so that one thread does not wait
// for each of them in turn
SPMD_HD inline double sum_in_order(double t, const double *p, int count)
{
    int j = 0;
    for (; j + 16 <= count; j += 16) {
        double v[16];
        SPMD_UNROLL
        for (int q = 0; q < 16; ++q) v[q] = p[j + q];
        SPMD_UNROLL
        for (int q = 0; q < 16; ++q) t = t + v[q];
    }
    for (; j + 4 <= count; j += 4) {
        const double p0 = p[j], p1 = p[j + 1], p2 = p[j + 2], p3 = p[j + 3];
        t = t + p0; t = t + p1; t = t + p2; t = t + p3;
    }
    for (; j < count; ++j) t = t + p[j];
    return t;
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
    bool dual_ready = false;           // w already holds the duals of the columns in Z (made by the Householder pass)
    const int itmax = 3 * ncol, iz2 = ncol, m = mrow;
    if (mrow <= 0 || ncol <= 0) return 2;
    PAR_FOR(T, i, 0, ncol) { index[i] = i + 1; x[i] = 0.0; }
    T.sync();
    for (;;) {
        // ---- L110: dual variables of the columns still in Z
        if (iz1 > iz2 || nsetp >= m) break;
        // two columns per thread at a time: twice as many loads in flight from the matrix (global memory), each
        // column's sum still in the order of its rows.  (Not after a Householder pass that left the set as it is: that
        // pass has formed these very sums from the entries it was writing, see below.)
        for (int t = iz1 - 1 + T.tid(); t < iz2 && !dual_ready; t += 2 * T.size()) {
            const int t2 = t + T.size();
            const bool two = t2 < iz2;
            const int j = index[t], j2 = two ? index[t2] : j;
            double s = 0.0, s2 = 0.0;
            // a * (+-0.0) added to a sum that is never -0.0 changes nothing: rows where b is zero are not read.
            // Sixteen rows at a time, loads first, so that a thread does not wait for each load in turn.
            int r = npp1;
            for (; r + 15 <= m; r += 16) {
                double bv[16], av[16], av2[16];
                SPMD_UNROLL
                for (int q = 0; q < 16; ++q) bv[q] = b[r - 1 + q];
                SPMD_UNROLL
                for (int q = 0; q < 16; ++q) {
                    av[q] = bv[q] != 0.0 ? AT(r + q, j) : 0.0;
                    av2[q] = bv[q] != 0.0 ? AT(r + q, j2) : 0.0;
                }
                SPMD_UNROLL
                for (int q = 0; q < 16; ++q)
                    if (bv[q] != 0.0) { s = s + av[q] * bv[q]; s2 = s2 + av2[q] * bv[q]; }
            }
            for (; r <= m; ++r) {
                const double br = b[r - 1];
                if (br != 0.0) { s = s + AT(r, j) * br; s2 = s2 + AT(r, j2) * br; }
            }
            w[j - 1] = s;
            if (two) w[j2 - 1] = s2;
        }
        dual_ready = false;
        T.sync();
        QP_MARK(T, 16);   // NNLS: dual pass
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
                // tree over the team's partial results: the larger value wins, the earlier position on a tie (what a
                // scan from the left keeps), so the outcome does not depend on the team size
                int width = 1;
                while (width < T.size()) width *= 2;
                for (int half = width / 2; half > 0; half /= 2) {
                    const int q = T.tid() + half;
                    if (T.tid() < half && q < T.size()) {
                        const double v = W.redv[q];
                        const int p = W.redi[q];
                        const double v0 = W.redv[T.tid()];
                        const int p0 = W.redi[T.tid()];
                        if (p != 0 && (p0 == 0 || v > v0 || (v == v0 && p < p0))) { W.redv[T.tid()] = v; W.redi[T.tid()] = p; }
                    }
                    T.sync();
                }
                if (T.tid() == 0) {
                    W.sc[0] = W.redi[0] != 0 ? W.redv[0] : 0.0;
                    W.isc[0] = W.redi[0];
                }
                T.sync();
            }
            if (W.sc[0] <= 0.0) { accepted = -1; break; }      // exit loop A
            iz = W.isc[0];
            j = index[iz - 1];
            // ---- step four: test column j for linear dependence (thread 0, on a copy of the column: only the
            //      new pivot entry is written back, and only when the column is accepted)
            PAR_FOR(T, r, 0, m) { W.piv[r] = AT(r + 1, j); z[r] = b[r]; }
            T.sync();
            bool tested = false;
#if defined(__HIP_DEVICE_COMPILE__)
            if constexpr (Team::kWave) {
                if (m <= lsqwave::kMaxN) {
                    // The same test on the team's first wavefront (lsq_wave.hpp): the two ordered sums over the column
                    // run as chains of DPP additions (entry i of the column in slot i + 128 - m, so that every sum ends
                    // at the last slot and is entered at its first term), the largest magnitude by an exchange inside
                    // the rows of 16 lanes (a maximum does not depend on the order it is formed in; NaN entries lose
                    // every comparison in both forms), the update of z by the lanes.  Scalars are computed by every
                    // lane alike; lane 0 files them away.
                    tested = true;
                    if (T.tid() < 64) {
                        const int lane = lsqwave::lane_here(), c16 = lane & 15, pad = lsqwave::kMaxN - m;
                        double pcv[8], zv[8];
                        SPMD_UNROLL
                        for (int rr = 0; rr < 8; ++rr) {
                            const int i = 16 * rr + c16 - pad;
                            pcv[rr] = i >= 0 ? W.piv[i] : 0.0;
                            zv[rr] = i >= 0 ? z[i] : 0.0;
                        }
                        const double p0 = W.piv[npp1 - 1], z0 = z[npp1 - 1];
                        double cl = fabs(p0), upv = 0.0, pivot = p0, znew = z0;
                        int ok = 1;
                        if (npp1 + 1 <= m) {               // h12 does nothing when l1 > m
                            SPMD_UNROLL
                            for (int rr = 0; rr < 8; ++rr) {
                                const int i = 16 * rr + c16 - pad;
                                const double sm = fabs(pcv[rr]);
                                if (i >= npp1) cl = sm > cl ? sm : cl;
                            }
                            SPMD_UNROLL
                            for (int off = 8; off > 0; off >>= 1) {
                                const double o = __shfl_xor(cl, off, 16);
                                cl = o > cl ? o : cl;
                            }
                            if (cl > 0.0) {
                                const double clinv = 1.0 / cl;
                                double sm = (p0 * clinv) * (p0 * clinv);
                                double P[8];
                                SPMD_UNROLL
                                for (int rr = 0; rr < 8; ++rr) P[rr] = (pcv[rr] * clinv) * (pcv[rr] * clinv);
                                sm = lsqwave::ordered_sum_from(sm, P, pad + npp1);
                                cl = cl * sqrt(sm);
                                if (p0 > 0.0) cl = -cl;
                                upv = p0 - cl;
                                pivot = cl;
                            }
                        }
                        {
                            const double unorm = nrm2(nsetp, W.piv, 1);      // the rows already in the triangle
                            const double t = factor * fabs(pivot);
                            volatile double d = unorm + t;
                            if (d - unorm <= 0.0) ok = 0;
                        }
                        if (ok) {
                            // h12 apply (mode 2) to z (= b) with pivot npp1, l1 = npp1 + 1
                            if (npp1 + 1 <= m && fabs(pivot) > 0.0) {
                                double bq = upv * pivot;
                                if (bq < 0.0) {
                                    bq = 1.0 / bq;
                                    double sm = z0 * upv;
                                    double P[8];
                                    SPMD_UNROLL
                                    for (int rr = 0; rr < 8; ++rr) P[rr] = zv[rr] * pcv[rr];
                                    sm = lsqwave::ordered_sum_from(sm, P, pad + npp1);
                                    if (sm != 0.0) {
                                        sm = sm * bq;
                                        znew = z0 + sm * upv;
                                        if (lane < 16) {
                                            SPMD_UNROLL
                                            for (int rr = 0; rr < 8; ++rr) {
                                                const int i = 16 * rr + c16 - pad;
                                                if (i >= npp1) z[i] = zv[rr] + sm * pcv[rr];
                                            }
                                        }
                                    }
                                }
                            }
                            if (!(znew / pivot > 0.0)) ok = 0;
                        }
                        if (lane == 0) {
                            W.piv[npp1 - 1] = pivot;
                            z[npp1 - 1] = znew;
                            if (ok) AT(npp1, j) = pivot;
                            else w[j - 1] = 0.0;
                            W.isc[1] = ok;
                            W.sc[1] = upv;
                        }
                    }
                }
            }
#endif
            if (!tested && T.tid() == 0) {
                double *pc = W.piv;                  // pc[r - 1] = a(r, j)
                // h12 construct on rows npp1..m of column j
                double cl = fabs(pc[npp1 - 1]), upv = 0.0;
                int ok = 1;
                if (npp1 + 1 <= m) {               // h12 does nothing when l1 > m
                    for (int r = npp1 + 1; r <= m; ++r) { const double sm = fabs(pc[r - 1]); cl = sm > cl ? sm : cl; }
                    if (cl > 0.0) {
                        const double clinv = 1.0 / cl;
                        double sm = (pc[npp1 - 1] * clinv) * (pc[npp1 - 1] * clinv);
                        for (int r = npp1 + 1; r <= m; ++r) sm = sm + (pc[r - 1] * clinv) * (pc[r - 1] * clinv);
                        cl = cl * sqrt(sm);
                        if (pc[npp1 - 1] > 0.0) cl = -cl;
                        upv = pc[npp1 - 1] - cl;
                        pc[npp1 - 1] = cl;
                    }
                }
                {
                    const double unorm = nrm2(nsetp, pc, 1);         // the rows already in the triangle
                    const double t = factor * fabs(pc[npp1 - 1]);
                    volatile double d = unorm + t;
                    if (d - unorm <= 0.0) ok = 0;
                }
                if (ok) {
                    // h12 apply (mode 2) to z (= b) with pivot npp1, l1 = npp1 + 1
                    if (npp1 + 1 <= m && fabs(pc[npp1 - 1]) > 0.0) {
                        double bq = upv * pc[npp1 - 1];
                        if (bq < 0.0) {
                            bq = 1.0 / bq;
                            double sm = z[npp1 - 1] * upv;
                            for (int r = npp1 + 1; r <= m; ++r) sm = sm + z[r - 1] * pc[r - 1];
                            if (sm != 0.0) {
                                sm = sm * bq;
                                z[npp1 - 1] = z[npp1 - 1] + sm * upv;
                                for (int r = npp1 + 1; r <= m; ++r) z[r - 1] = z[r - 1] + sm * pc[r - 1];
                            }
                        }
                    }
                    if (!(z[npp1 - 1] / pc[npp1 - 1] > 0.0)) ok = 0;
                }
                if (ok) AT(npp1, j) = pc[npp1 - 1];
                else w[j - 1] = 0.0;
                W.isc[1] = ok;
                W.sc[1] = upv;
            }
            T.sync();
            if (W.isc[1]) { accepted = 1; up = W.sc[1]; break; }
        }
        QP_MARK(T, 17);   // NNLS: candidate search and test
        if (accepted < 0) break;
        // ---- step five: column j enters the set
        PAR_FOR(T, r, 0, m) b[r] = z[r];                   // W.piv already holds the column with its new pivot
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
            // h12 apply (pivot nsetp, l1 = npp1) to every column still in Z.  The same walk forms the column's dual for
            // the next pass of the loop, w = sum over rows npp1..m of a(r, jj) b(r) in the order of the rows (b is final
            // since step five; rows where b is zero are left out as in the dual pass): from the entries as they are
            // read if the column stays as it is, from the new entries as they are written otherwise.
            // Two columns per thread at a time (twice as many loads in flight: the walk waits for the matrix in global
            // memory), each column's sums still in the order of its rows.  (Tried and slower on the device: the next
            // block of sixteen rows requested before this one is summed, and the last partial block as a masked block
            // instead of a row-by-row tail.)
            const double pv = W.piv[nsetp - 1];
            const int applies = (npp1 <= m) && (fabs(pv) > 0.0);
            double bq = up * pv;
            if (applies && bq < 0.0) {
                bq = 1.0 / bq;
                dual_ready = true;
                for (int t = iz1 - 1 + T.tid(); t < iz2; t += 2 * T.size()) {
                    const int t2 = t + T.size();
                    const bool two = t2 < iz2;
                    const int jA = index[t], jB = two ? index[t2] : jA;
                    const double a0A = AT(nsetp, jA), a0B = AT(nsetp, jB);
                    double smA = a0A * up, smB = a0B * up, dA = 0.0, dB = 0.0;
                    int r = npp1;
                    for (; r + 15 <= m; r += 16) {
                        double vA[16], vB[16];
                        SPMD_UNROLL
                        for (int q = 0; q < 16; ++q) { vA[q] = AT(r + q, jA); vB[q] = AT(r + q, jB); }
                        SPMD_UNROLL
                        for (int q = 0; q < 16; ++q) {
                            const double pq = W.piv[r - 1 + q], br = b[r - 1 + q];
                            smA = smA + vA[q] * pq;
                            smB = smB + vB[q] * pq;
                            if (br != 0.0) { dA = dA + vA[q] * br; dB = dB + vB[q] * br; }
                        }
                    }
                    for (; r <= m; ++r) {
                        const double pq = W.piv[r - 1], br = b[r - 1], vA = AT(r, jA), vB = AT(r, jB);
                        smA = smA + vA * pq;
                        smB = smB + vB * pq;
                        if (br != 0.0) { dA = dA + vA * br; dB = dB + vB * br; }
                    }
                    // the update of a column whose sum is not zero, and its dual from the new entries
                    auto update = [&](int jj, double a0, double sm) -> double {
                        sm = sm * bq;
                        AT(nsetp, jj) = a0 + sm * up;
                        double d = 0.0;
                        int r = npp1;
                        for (; r + 15 <= m; r += 16) {
                            double v[16];
                            SPMD_UNROLL
                            for (int q = 0; q < 16; ++q) v[q] = AT(r + q, jj);
                            SPMD_UNROLL
                            for (int q = 0; q < 16; ++q) {
                                const double an = v[q] + sm * W.piv[r - 1 + q], br = b[r - 1 + q];
                                AT(r + q, jj) = an;
                                if (br != 0.0) d = d + an * br;
                            }
                        }
                        for (; r <= m; ++r) {
                            const double an = AT(r, jj) + sm * W.piv[r - 1], br = b[r - 1];
                            AT(r, jj) = an;
                            if (br != 0.0) d = d + an * br;
                        }
                        return d;
                    };
                    if (two && smA != 0.0 && smB != 0.0) {          // the usual case: both columns together again
                        smA = smA * bq;
                        smB = smB * bq;
                        AT(nsetp, jA) = a0A + smA * up;
                        AT(nsetp, jB) = a0B + smB * up;
                        dA = 0.0; dB = 0.0;
                        r = npp1;
                        for (; r + 15 <= m; r += 16) {
                            double vA[16], vB[16];
                            SPMD_UNROLL
                            for (int q = 0; q < 16; ++q) { vA[q] = AT(r + q, jA); vB[q] = AT(r + q, jB); }
                            SPMD_UNROLL
                            for (int q = 0; q < 16; ++q) {
                                const double pq = W.piv[r - 1 + q], br = b[r - 1 + q];
                                const double nA = vA[q] + smA * pq, nB = vB[q] + smB * pq;
                                AT(r + q, jA) = nA;
                                AT(r + q, jB) = nB;
                                if (br != 0.0) { dA = dA + nA * br; dB = dB + nB * br; }
                            }
                        }
                        for (; r <= m; ++r) {
                            const double pq = W.piv[r - 1], br = b[r - 1];
                            const double nA = AT(r, jA) + smA * pq, nB = AT(r, jB) + smB * pq;
                            AT(r, jA) = nA;
                            AT(r, jB) = nB;
                            if (br != 0.0) { dA = dA + nA * br; dB = dB + nB * br; }
                        }
                    } else {
                        if (smA != 0.0) dA = update(jA, a0A, smA);
                        if (two && smB != 0.0) dB = update(jB, a0B, smB);
                    }
                    w[jA - 1] = dA;
                    if (two) w[jB - 1] = dB;
                }
            }
        }
        PAR_FOR(T, r, npp1, m + 1) AT(r, j) = 0.0;         // rows npp1..m of the new column
        if (T.tid() == 0) w[j - 1] = 0.0;
        T.sync();
        QP_MARK(T, 18);   // NNLS: Householder application
        // ---- steps six to eleven: solve, step length, possibly drop columns.  The arithmetic is the serial text's,
        //      entry for entry; what runs in parallel are the entries that do not depend on each other: the rows of one
        //      elimination step of the triangular solve (on the team's first wavefront, whose LDS traffic needs no
        //      workgroup barrier) and the columns of one Givens rotation (whole team).
        int failed = 0;
        for (;;) {
            // L180: solve the triangular system for z.  Row r takes the term of row k > r in the order k = nsetp,
            // nsetp - 1, ..., and is divided by its pivot once row r + 1 is final: one step per row, the rows above it
            // in parallel.  The pivots are fetched first, all at once, and the columns a panel of eight steps needs are
            // fetched together before its first step (the matrix lives in global memory: one latency per panel).
            const int wl = T.size() < 64 ? T.size() : 64;
            if (T.tid() < wl) {
                for (int ip = 1 + T.tid(); ip <= nsetp; ip += wl) W.piv[ip - 1] = AT(ip, index[ip - 1]);
                T.wave_sync();
                constexpr int P = 8;
                const int r0 = 1 + T.tid(), r1 = r0 + wl;
                for (int top = nsetp; top >= 1; top -= P) {
                    double v0[P], v1[P];
                    SPMD_UNROLL
                    for (int q = 0; q < P; ++q) {
                        const int ip = top - q;
                        v0[q] = 0.0; v1[q] = 0.0;
                        if (ip >= 1 && ip != nsetp) {
                            const int col = index[ip];
                            if (r0 <= ip) v0[q] = AT(r0, col);
                            if (r1 <= ip) v1[q] = AT(r1, col);
                        }
                    }
                    SPMD_UNROLL
                    for (int q = 0; q < P; ++q) {
                        const int ip = top - q;
                        if (ip >= 1) {
                            if (ip != nsetp) {
                                const double zf = -z[ip];
                                if (zf != 0.0) {
                                    if (r0 <= ip) z[r0 - 1] = z[r0 - 1] + zf * v0[q];
                                    if (r1 <= ip) z[r1 - 1] = z[r1 - 1] + zf * v1[q];
                                    const int col = index[ip];
                                    for (int r = r1 + wl; r <= ip; r += wl) z[r - 1] = z[r - 1] + zf * AT(r, col);
                                }
                                T.wave_sync();
                            }
                            if (T.tid() == 0) z[ip - 1] = z[ip - 1] / W.piv[ip - 1];
                            T.wave_sync();
                        }
                    }
                }
                if (T.tid() == 0) {
                    int status = 0, jj = 0;      // 0: back to L110, 1: drop columns starting at position jj, 3: give up
                    iter = iter + 1;
                    if (iter > itmax) status = 3;
                    else {
                        double alpha = 1.0;
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
                        status = jj == 0 ? 0 : 1;
                    }
                    W.isc[2] = status;
                    W.isc[3] = iter;
                    W.isc[7] = jj;
                }
            }
            T.sync();
            iter = W.isc[3];
            if (W.isc[2] == 3) { failed = 1; break; }
            if (W.isc[2] == 0) break;                         // back to L110
            dual_ready = false;                               // columns leave the set: rows are rotated, the duals are stale
            int jj = W.isc[7];
            int i = index[jj - 1];
            int emptied = 0;
            for (;;) {
                // L250: move column i out of the set, restore the triangle with Givens rotations
                T.sync();                                     // everyone has read the scalars of the previous pass
                if (T.tid() == 0) x[i - 1] = 0.0;
                jj = jj + 1;
                for (int jc = jj; jc <= nsetp; ++jc) {
                    const int ii = index[jc - 1];
                    if (T.tid() == 0) {
                        index[jc - 2] = ii;
                        double da = AT(jc - 1, ii), db = AT(jc, ii), c, s;
                        {
                            double roe = db, r;
                            if (fabs(da) > fabs(db)) roe = da;
                            const double scale = fabs(da) + fabs(db);
                            if (scale == 0.0) { c = 1.0; s = 0.0; r = 0.0; }
                            else {
                                r = scale * sqrt((da / scale) * (da / scale) + (db / scale) * (db / scale));
                                r = (roe >= 0.0 ? 1.0 : -1.0) * r;
                                c = da / r;
                                s = db / r;
                            }
                            da = r;
                        }
                        W.sc[8] = c; W.sc[9] = s; W.sc[10] = da;
                        const double dx = b[jc - 2], dy = b[jc - 1];
                        const double tt = c * dx + s * dy;
                        b[jc - 1] = c * dy - s * dx;
                        b[jc - 2] = tt;
                    }
                    T.sync();
                    const double c = W.sc[8], s = W.sc[9];
                    PAR_FOR(T, q, 1, ncol + 1) {              // rows jc-1 and jc of every column
                        if (q == ii) {                        // the rotated pair itself: (r, 0)
                            AT(jc - 1, q) = W.sc[10];
                            AT(jc, q) = 0.0;
                        } else {
                            const double dx = AT(jc - 1, q), dy = AT(jc, q);
                            const double tt = c * dx + s * dy;
                            AT(jc, q) = c * dy - s * dx;
                            AT(jc - 1, q) = tt;
                        }
                    }
                    T.sync();
                }
                npp1 = nsetp;
                nsetp = nsetp - 1;
                iz1 = iz1 - 1;
                if (T.tid() == 0) {
                    index[iz1 - 1] = i;
                    int flag = 0, pos = 0, col = 0;           // 0: feasible again, 1: drop the column at pos, 2: set is empty
                    if (nsetp <= 0) flag = 2;
                    else {
                        for (pos = 1; pos <= nsetp; ++pos) {
                            col = index[pos - 1];
                            if (x[col - 1] <= 0.0) { flag = 1; break; }
                        }
                    }
                    W.isc[12] = flag; W.isc[7] = pos; W.isc[13] = col;
                }
                T.sync();
                if (W.isc[12] == 2) { emptied = 1; break; }
                if (W.isc[12] == 0) break;
                jj = W.isc[7];
                i = W.isc[13];
            }
            if (emptied) { failed = 1; break; }
            PAR_FOR(T, r, 0, m) z[r] = b[r];
            T.sync();
        }
        QP_MARK(T, 19);   // NNLS: steps six to eleven
        if (failed) { mode = 3; break; }
    }
    // ---- L280
    if (T.tid() == 0) {
        const int k = npp1 < m ? npp1 : m;
        *rnorm_out = nrm2(m - nsetp, &b[k - 1], 1);
        W.isc[11] = nsetp;                                 // index[0..nsetp) is the final set
    }
    if (npp1 > m) PAR_FOR(T, i, 0, ncol) w[i] = 0.0;
    T.sync();
    return mode;
#undef AT
}

// Rows of [g h] <- [g e^-1, h - g f] for the MPC's constraint matrix, R rows per call, each written to
// dst[r][j * mg], j = 0..n (entry n is h).  A row is generated from the structured Jacobian / the bounds; columns are
// taken NB * 6 at a time so that an earlier entry of the row is read back once per block, the R rows share every read
// of e, and every sum runs over k in ascending order as in the serial text.  Blocks of earlier columns in which none
// of the R rows has a nonzero yet are left out of the sums: their entries are +-0.0, and +-0.0 * e added to a sum
// that is +0.0 (or to any nonzero sum) changes nothing.
#ifndef MPC_QP_NB
#define MPC_QP_NB 2
#endif
#ifndef MPC_QP_FREE_STEP
#define MPC_QP_FREE_STEP 1         // lsq tries the unconstrained step first (0: always the long form; development)
#endif
constexpr double kFreeStepMargin = 1e-6;    // a row passes if its slack is at least this times the magnitude of its terms
constexpr double kFreeStepSpread = 1e-3;    // ... and e's diagonal entries lie within 1e-3 .. 1e3 of the first
#ifndef MPC_QP_ROWS
#define MPC_QP_ROWS 1              // rows per lane; see lsq's row loop for why one beats two on the device
#endif
#ifndef MPC_QP_SNAKE
#define MPC_QP_SNAKE 1
#endif

struct RowDesc {
    int a, first, last, kind, kk;      // sub-blocks [first, last] hold the row's nonzeros, in column 6 * jj + a of each
    double hval;
    double hval_twin;                  // kind 3 (lower bound of variable v): h of the upper-bound row of v, -hi[v]
};

// The upper-bound row of a variable is the lower-bound row's negative, entry for entry: the row is -e_v instead of
// +e_v, every product of the triangular solve changes its sign exactly, sums of terms that all changed sign round to
// the negative of the sum, and a sum or difference that is zero is +0.0 in both rows (it starts from +0.0, and
// x - x = +0.0).  So the "twin" is written as 0.0 - (the lower-bound row's entry) -- -x for a nonzero x, +0.0 for
// +0.0 -- and never computed: a third of the rows that begin late, a sixth of the transform's work.  The twins' h
// entries are written with their rows' (they need the row's sum against f, which is not stored); the other n columns
// by one pass of the whole team after the transform (contiguous loads and stores, 960 bytes per column).
SPMD_HD inline double twin_of(double v) { return 0.0 - v; }

// aug: the rows are numbered as in Kraft's augmented problem (one more variable, delta, whose two bound rows follow
// the lower and the upper bounds of the others: m constraint rows, n + 1 lower bounds, n + 1 upper bounds); delta's own
// rows are not described here (delta_row below writes them).
SPMD_HD inline RowDesc describe_row(int H, int i, const double *cvals, const double *lo, const double *hi, int aug = 0)
{
    const int n = 6 * H, m = 12 * (H + 1);
    RowDesc d;
    d.a = 0; d.first = H; d.last = -1; d.kind = 0; d.kk = 0; d.hval_twin = 0.0;
    if (i < m) {
        const int st = i / 12, r = i % 12;
        d.hval = -cvals[i];
        if (r < 6) { d.a = r; d.kk = st; if (st >= 1) { d.first = 0; d.last = st - 1; d.kind = 1; } }        // position row
        else { d.a = r - 6; if (st >= 1) { d.first = st - 1; d.last = st - 1; d.kind = 2; } }                // velocity row
    } else if (i < m + n) {
        const int v = i - m;
        d.a = v % 6; d.first = v / 6; d.last = d.first; d.kind = 3; d.hval = lo[v]; d.hval_twin = -hi[v];
    } else {
        const int v = i - m - n - aug;
        d.a = v % 6; d.first = v / 6; d.last = d.first; d.kind = 4; d.hval = -hi[v];
    }
    return d;
}

#if defined(__HIP_DEVICE_COMPILE__)
// The double that lane L of each row of 16 lanes holds in v, for every lane of that row: one v_mov_b64 with the DPP
// row broadcast.  (Through the scalar unit - two v_readlane - the multiply that uses it waits ~20 cycles longer.)
template <int L>
__device__ __forceinline__ double row_bcast(double v)
{
    return __builtin_amdgcn_update_dpp(0.0, v, 0x150 + L, 0xf, 0xf, true);
}
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// transform_cols for a whole wavefront (every lane is here, control flow is uniform; a lane's own rows are rows[], row i
// of the result at out[i + j mg]: one scalar base per column and one lane offset per row address every entry).
// Entries of e are not read by every lane from LDS (64 lanes reading one address cost the LDS pipe 64 lanes of
// bandwidth, and one LDS instruction costs a lone wavefront ~30 cycles of issue): the 6 x NC block that six steps need
// is loaded once, entry t = 16 j + lane % 16 into register j of every row of 16 lanes, and each entry reaches the
// multiplies by row_bcast.  The same for the triangle inside the column block and for f.  The row's earlier entries
// (global memory) and the next block of e are requested one block ahead.
// A lane whose rows start later than the wavefront's first block (kb_lo) walks the earlier blocks with its rows'
// stored entries, (0 - 0) / e(j, j): what the serial text does with a row's leading zeros.
template <int NB, int R>
__device__ __forceinline__ void transform_cols_wave(const Work &W, int n, int mg, int H, const double *nz, double *out,
                                                    const int *rows, int jj0, const RowDesc *d, double *th, int kb_lo, bool store)
{
    constexpr int NC = 6 * NB, NE = 6 * NC, NR = (NE + 15) / 16, NT = NC * (NC + 1) / 2, NTR = (NT + 15) / 16;
    const int jb = 6 * jj0, lane16 = (int)(threadIdx.x & 15u);
    unsigned rowb[R];                                      // a row's place in a column, in bytes: the lane's 32-bit offset
    SPMD_UNROLL
    for (int r = 0; r < R; ++r) rowb[r] = 8u * (unsigned)rows[r];
#define EO(r) ((r) * n - ((r) * ((r) - 1)) / 2)
    double acc[R][NC];
    SPMD_UNROLL
    for (int r = 0; r < R; ++r) {
        SPMD_UNROLL
        for (int c = 0; c < NC; ++c) acc[r][c] = 0.0;
    }
    if (kb_lo < jj0) {
        // entry t of block kb is e(6 kb + t / NC, jb + t % NC), at E[f(k) + jb + t % NC] with k = 6 kb + t / NC and
        // f(k) = k (n - 1) - k (k - 1) / 2; from one block to the next f grows by 6 (n - 1) - 6 k - 15
        int off[NR], step[NR];
        SPMD_UNROLL
        for (int j = 0; j < NR; ++j) {
            const int t = 16 * j + lane16 < NE ? 16 * j + lane16 : 0;
            const int k = 6 * kb_lo + t / NC;
            off[j] = k * (n - 1) - (k * (k - 1)) / 2 + jb + t % NC;
            step[j] = 6 * (n - 1) - 6 * k - 15;
        }
        // Block kb + 1 (its entries of e, the rows' own earlier entries from global memory) is requested before block
        // kb is used; two register sets take turns, the loop is unrolled by two so that neither is ever copied.
        double ea[NR], eb[NR], ga[R][6], gb[R][6];
        auto fetch = [&](int kb, double (&e)[NR], double (&g)[R][6]) {
            SPMD_UNROLL
            for (int j = 0; j < NR; ++j) e[j] = W.E[off[j]];
            SPMD_UNROLL
            for (int r = 0; r < R; ++r) {
                SPMD_UNROLL
                for (int q = 0; q < 6; ++q) g[r][q] = *(const double *)((const char *)(out + (size_t)(6 * kb + q) * mg) + rowb[r]);
            }
            SPMD_UNROLL
            for (int j = 0; j < NR; ++j) { off[j] += step[j]; step[j] -= 36; }
        };
        auto block = [&](int kb, const double (&e)[NR], const double (&g)[R][6], double (&en)[NR], double (&gn)[R][6]) {
            if (kb + 1 < jj0) fetch(kb + 1, en, gn);
            // One entry at a time, in this order: each entry's sums pass through an empty volatile asm, which instruction
            // selection and the scheduler keep in place (left alone they issue all the broadcasts of a block first, and
            // their 72 results spill; a scheduling barrier alone comes too late, the order is chosen before it acts).
            static_for<0, NE>([&](auto T) {
                constexpr int t = decltype(T)::value, q = t / NC, c = t % NC;
                const double b = row_bcast<t % 16>(e[t / 16]);
                SPMD_UNROLL
                for (int r = 0; r < R; ++r) {
                    acc[r][c] = acc[r][c] + g[r][q] * b;
                    asm volatile("" : "+v"(acc[r][c]));
                }
            });
        };
        fetch(kb_lo, ea, ga);
        int kb = kb_lo;
        for (; kb + 1 < jj0; kb += 2) {
            block(kb, ea, ga, eb, gb);
            block(kb + 1, eb, gb, ea, ga);
        }
        if (kb < jj0) block(kb, ea, ga, eb, gb);
    }
    // the triangle inside the column block, entry u = c (c + 1) / 2 + c2 = e(jb + c2, jb + c), c2 <= c, and f
    double tri[NTR];
    SPMD_UNROLL
    for (int j = 0; j < NTR; ++j) {
        const int u = 16 * j + lane16 < NT ? 16 * j + lane16 : 0;
        int c = 0;
        SPMD_UNROLL
        for (int cc = 1; cc < NC; ++cc) c += u >= (cc * (cc + 1)) / 2 ? 1 : 0;
        const int c2 = u - (c * (c + 1)) / 2;
        tri[j] = W.E[EO(jb + c2) + (c - c2)];
    }
    double fv[(NC + 15) / 16];                             // f(jb + t) in lane t % 16 of register t / 16
    SPMD_UNROLL
    for (int j = 0; j < (NC + 15) / 16; ++j) fv[j] = W.ff[jb + (16 * j + lane16 < NC ? 16 * j + lane16 : 0)];
    double val[R][NB];
    SPMD_UNROLL
    for (int r = 0; r < R; ++r) {
        SPMD_UNROLL
        for (int b = 0; b < NB; ++b) {
            const int jj = jj0 + b;
            double v = 0.0;
            if (jj >= d[r].first && jj <= d[r].last) {
                if (d[r].kind == 1) v = nz[6 * H + 6 * (jj * H - (jj * (jj - 1)) / 2 - jj - 1 + d[r].kk) + d[r].a];
                else if (d[r].kind == 2) v = nz[6 * jj + d[r].a];
                else v = (d[r].kind == 3) ? 1.0 : -1.0;
            }
            val[r][b] = v;
        }
    }
    static_for<0, NC>([&](auto C) {
        constexpr int c = decltype(C)::value;
        double t[R];
        SPMD_UNROLL
        for (int r = 0; r < R; ++r) t[r] = acc[r][c];
        static_for<0, c>([&](auto C2) {
            constexpr int c2 = decltype(C2)::value, u = (c * (c + 1)) / 2 + c2;
            const double e = row_bcast<u % 16>(tri[u / 16]);
            SPMD_UNROLL
            for (int r = 0; r < R; ++r) {
                t[r] = t[r] + acc[r][c2] * e;                                 // acc[r][c2] holds the new g(i_r, jb + c2)
                asm volatile("" : "+v"(t[r]));
            }
        });
        constexpr int ud = (c * (c + 1)) / 2 + c;
        const double diag = row_bcast<ud % 16>(tri[ud / 16]), fj = row_bcast<c % 16>(fv[c / 16]);
        SPMD_UNROLL
        for (int r = 0; r < R; ++r) {
            const double orig = (c % 6 == d[r].a) ? val[r][c / 6] : 0.0;
            const double gn = (orig - t[r]) / diag;
            acc[r][c] = gn;
            if (store) *(double *)((char *)(out + (size_t)(jb + c) * mg) + rowb[r]) = gn;
            th[r] = th[r] + gn * fj;
        }
    });
#undef EO
}
#endif

template <int NB, int R>
SPMD_HD inline void transform_cols(const Work &W, int n, int mg, int H, const double *nz, double *const *dst, int jj0,
                                   const RowDesc *d, int first_any, double *th, bool store = true)
{
    constexpr int NC = 6 * NB;
    const int jb = 6 * jj0;
#define EO(r) ((r) * n - ((r) * ((r) - 1)) / 2)
    double acc[R][NC];
    SPMD_UNROLL
    for (int r = 0; r < R; ++r) {
        SPMD_UNROLL
        for (int c = 0; c < NC; ++c) acc[r][c] = 0.0;
    }
    {
        const double *er = W.E + jb;                       // er[c] = e(k, jb + c), advanced row by row
        for (int kb = 0; kb < jj0; ++kb) {
            if (kb >= first_any) {
                double gk[R][6];
                SPMD_UNROLL
                for (int r = 0; r < R; ++r) {
                    SPMD_UNROLL
                    for (int q = 0; q < 6; ++q) gk[r][q] = dst[r][(size_t)(6 * kb + q) * mg];
                }
                SPMD_UNROLL
                for (int q = 0; q < 6; ++q) {
                    double e[NC];                          // the row of e first, in one go: one wait instead of NC
                    SPMD_UNROLL
                    for (int c = 0; c < NC; ++c) e[c] = er[c];
                    SPMD_UNROLL
                    for (int c = 0; c < NC; ++c) {
                        SPMD_UNROLL
                        for (int r = 0; r < R; ++r) acc[r][c] = acc[r][c] + gk[r][q] * e[c];
                    }
                    er += n - (6 * kb + q) - 1;
                }
            } else {
                er += 6 * (n - 6 * kb - 1) - 15;
            }
        }
    }
    double val[R][NB];
    SPMD_UNROLL
    for (int r = 0; r < R; ++r) {
        SPMD_UNROLL
        for (int b = 0; b < NB; ++b) {
            const int jj = jj0 + b;
            double v = 0.0;
            if (jj >= d[r].first && jj <= d[r].last) {
                if (d[r].kind == 1) v = nz[6 * H + 6 * (jj * H - (jj * (jj - 1)) / 2 - jj - 1 + d[r].kk) + d[r].a];
                else if (d[r].kind == 2) v = nz[6 * jj + d[r].a];
                else v = (d[r].kind == 3) ? 1.0 : -1.0;
            }
            val[r][b] = v;
        }
    }
    SPMD_UNROLL
    for (int c = 0; c < NC; ++c) {
        double t[R];
        SPMD_UNROLL
        for (int r = 0; r < R; ++r) t[r] = acc[r][c];
        SPMD_UNROLL
        for (int c2 = 0; c2 < c; ++c2) {
            const double e = W.E[EO(jb + c2) + (c - c2)];
            SPMD_UNROLL
            for (int r = 0; r < R; ++r) t[r] = t[r] + acc[r][c2] * e;      // acc[r][c2] holds the new g(i_r, jb + c2)
        }
        const double diag = W.E[EO(jb + c)], fj = W.ff[jb + c];
        SPMD_UNROLL
        for (int r = 0; r < R; ++r) {
            const double orig = (c % 6 == d[r].a) ? val[r][c / 6] : 0.0;
            const double gn = (orig - t[r]) / diag;
            acc[r][c] = gn;
            if (store) dst[r][(size_t)(jb + c) * mg] = gn;
            th[r] = th[r] + gn * fj;
        }
    }
#undef EO
}

// Row number p of the m + n rows of [g h] that are computed (the upper-bound rows are written as their lower-bound
// rows' twins) in order of the work its transform takes: the 6H position rows of states 1 .. H (nonzeros from block 0
// on), then for f = 0 .. H - 1 the rows that begin at block f (the velocity rows of state f + 1, the lower-bound rows
// of its six variables), last the 12 rows of state 0 (no variable enters them).
SPMD_HD inline int work_order(int H, int p)
{
    const int n = 6 * H, m = 12 * (H + 1);
    if (p < n) return 12 * (p / 6 + 1) + p % 6;
    const int q = p - n;
    if (q < 12 * H) {
        const int f = q / 12, r = q % 12;
        if (r < 6) return 12 * (f + 1) + 6 + r;
        return m + 6 * f + (r - 6);
    }
    return q - 12 * H;
}

// AUG (Kraft's augmented problem, slsqp_core.c's lsq with n2 = 1): one more variable, delta, whose row of e is
// (0, ..., 0, rho) with f = -0.0.  A row of g gets one more entry, (a(i, n + 1) - 0) / rho with a(i, n + 1) =
// max(-c(i), 0) in a constraint row and 0 in a bound row: the sum that is subtracted runs over products with the
// zeros of e's last column and stays +0.0; the row's h is what it was, its last term g(i, n + 1) * (-0.0) leaves a sum
// that is never -0.0 as it is.  h moves one place down (column n + 1), rows are mg + 2 apart, the twins n + 1 further.
template <int R, bool COOP = false, bool AUG = false>
SPMD_HD inline void transform_rows(const Work &W, int H, const int *rows, const double *cvals, const double *nz,
                                   const double *lo, const double *hi, double *const *dst, bool store = true,
                                   double *base = nullptr,      // COOP: dst[r] = base + rows[r], base the same in every lane
                                   bool twin = false,           // also write the twin's h (row + n, column n) of every lower-bound row
                                   bool no_walk = false)        // e is diagonal (lsq's unit_e): no block before a column's own is read
{
    const int n = 6 * H, mg = 12 * (H + 1) + 2 * n + (AUG ? 2 : 0);        // mg: the distance between two columns
    RowDesc d[R];
    double th[R];
    int first_any = H;
    SPMD_UNROLL
    for (int r = 0; r < R; ++r) {
        d[r] = describe_row(H, rows[r], cvals, lo, hi, AUG ? 1 : 0);
        th[r] = 0.0;
        first_any = d[r].first < first_any ? d[r].first : first_any;
    }
    int skip_to = first_any;           // blocks before it are zeros for every row handled here
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (COOP) {              // ... for every row of the WAVEFRONT: its lanes walk the blocks together
        SPMD_UNROLL
        for (int off = 32; off > 0; off >>= 1) {
            const int other = __shfl_xor(skip_to, off);
            skip_to = other < skip_to ? other : skip_to;
        }
        skip_to = __builtin_amdgcn_readfirstlane(skip_to);   // the same in every lane: say so, the loops over blocks are scalar
    }
#endif
    int jj = 0;
    for (; jj + MPC_QP_NB <= H; jj += MPC_QP_NB) {
        if (jj + MPC_QP_NB <= skip_to) {                    // all zeros so far: (0 - 0) / e(j, j)
            for (int c = 0; c < 6 * MPC_QP_NB; ++c) {
                const int j = 6 * jj + c;
                const double res = (0.0 - 0.0) / W.E[j * n - (j * (j - 1)) / 2];
                if (store) {
                    SPMD_UNROLL
                    for (int r = 0; r < R; ++r) dst[r][(size_t)j * mg] = res;
                }
            }
        } else {
#if defined(__HIP_DEVICE_COMPILE__)
            if constexpr (COOP) transform_cols_wave<MPC_QP_NB, R>(W, n, mg, H, nz, base, rows, jj, d, th, no_walk ? jj : skip_to, store);
            else
#endif
                transform_cols<MPC_QP_NB, R>(W, n, mg, H, nz, dst, jj, d, no_walk ? jj : first_any, th, store);
        }
    }
    for (; jj < H; ++jj) {
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr (COOP) transform_cols_wave<1, R>(W, n, mg, H, nz, base, rows, jj, d, th, (no_walk || jj < skip_to) ? jj : skip_to, store);
        else
#endif
            transform_cols<1, R>(W, n, mg, H, nz, dst, jj, d, no_walk ? jj : first_any, th, store);
    }
    if (store) {
        SPMD_UNROLL
        for (int r = 0; r < R; ++r) {
            if constexpr (AUG) {
                const double last = (rows[r] < 12 * (H + 1) && d[r].hval > 0.0) ? d[r].hval : 0.0;     // max(-c, 0) / 0
                dst[r][(size_t)n * mg] = (last - 0.0) / W.E[(n * (n + 1)) / 2];
                dst[r][(size_t)(n + 1) * mg] = d[r].hval - th[r];
                if (twin && d[r].kind == 3) dst[r][(size_t)(n + 1) * mg + n + 1] = d[r].hval_twin - twin_of(th[r]);
            } else {
                dst[r][(size_t)n * mg] = d[r].hval - th[r];
                if (twin && d[r].kind == 3) dst[r][(size_t)n * mg + n] = d[r].hval_twin - twin_of(th[r]);
            }
        }
    }
}

// The two bound rows of the augmented problem's extra variable (0 <= delta <= 1): row `upper` ? -e_{n+1} : +e_{n+1}
// of g, transformed: zeros before the last entry, (0 - 0) / e(j, j) each; the last (+-1 - 0) / rho; h = 0 or -1 minus a sum
// of zeros that ends +0.0.  One thread writes a row (n + 2 entries, ld apart).
SPMD_HD inline void delta_row(const Work &W, int n, int ld, bool upper, double *dst)
{
    for (int j = 0; j < n; ++j) dst[(size_t)j * ld] = (0.0 - 0.0) / W.E[j * n - (j * (j - 1)) / 2];
    dst[(size_t)n * ld] = ((upper ? -1.0 : 1.0) - 0.0) / W.E[(n * (n + 1)) / 2];
    dst[(size_t)(n + 1) * ld] = (upper ? -1.0 : 0.0) - 0.0;
}

#if defined(__HIP_DEVICE_COMPILE__)
// transform_rows for a whole wavefront as a function of its own, not inlined: the kernel around it keeps ~140 vector
// registers alive through every phase, and inlined the transform has no room left to request a row's earlier entries
// one block ahead.  Arguments of a device function arrive in vector registers, uniform or not: the scalar ones are
// made scalar again (v_readfirstlane), and the two pool pointers come as LDS offsets when the pool is LDS, so that the
// reads of e stay LDS instructions.
__device__ __forceinline__ int scalar_again(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ unsigned long long scalar_bits(const void *p)
{
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
// ... and a pointer to global memory is said to be one: a pointer that arrives as an argument is generic otherwise, and a
// generic load needs its whole address in vector registers, where a global one takes a scalar base and a lane offset
__device__ __forceinline__ double *global_again(const double *p)
{
    typedef __attribute__((address_space(1))) double GlobalDouble;
    return (double *)(GlobalDouble *)scalar_bits(p);
}
template <int R, bool LDS>
__device__ __attribute__((noinline)) void transform_rows_call(const double *E, const double *ff, unsigned e_lds, unsigned ff_lds,
                                                              double *AT, int H, const double *cvals, const double *nz,
                                                              const double *lo, const double *hi, int row0, int row1, int store)
{
    static_assert(R == 1 || R == 2, "one or two rows per lane (row1 is ignored for one)");
    Work W{};
    if constexpr (LDS) {
        typedef __attribute__((address_space(3))) double LdsDouble;
        W.E = (double *)(LdsDouble *)(unsigned long long)scalar_again((int)e_lds);
        W.ff = (double *)(LdsDouble *)(unsigned long long)scalar_again((int)ff_lds);
    } else {
        W.E = global_again(E);
        W.ff = global_again(ff);
    }
    AT = global_again(AT);
    cvals = global_again(cvals);
    nz = global_again(nz);
    lo = global_again(lo);
    hi = global_again(hi);
    W.AT = AT;
    const int rows[2] = {row0, row1};
    double *const dst[2] = {AT + row0, AT + row1};
    if constexpr (R == 2) {
        transform_rows<2, true>(W, scalar_again(H), rows, cvals, nz, lo, hi, dst, store != 0, AT, true);
    } else {
        const int rows1[1] = {row0};
        double *const dst1[1] = {AT + row0};
        transform_rows<1, true>(W, scalar_again(H), rows1, cvals, nz, lo, hi, dst1, store != 0, AT, true);
    }
}
#endif

// Kraft's LSQ for the MPC shape.  W.E holds the packed LDL' factor l on entry ((n+1)n/2 + 1 entries; it is
// overwritten), g: gradient, cvals: c(x), nz: structured dc/dx, lo / hi: bounds on the step (xl - x, xu - x).
// Returns the mode (1 = solved) and writes the step s [n] and the multipliers y [m + 2n] (the bound rows'
// multipliers follow the m constraint rows; slsqp_core.c overwrites those with NaN, they are unused).
//
// Where the text below departs from a literal reading of slsqp_core.c, the arithmetic is still the same:
//   * rows of [g h] are generated from nz and the bounds instead of being read from a zero-filled matrix, and the
//     terms of a row's sums that come before its first nonzero are left out: they add (+-0) * e to a sum that is
//     +0.0, which leaves +0.0;
//   * sums against the NNLS solution skip its exact zeros (every entry outside the final set): t + a * 0.0 is t for
//     finite a, and t is never -0.0.
//
// AUG: Kraft's augmented problem for an inconsistent linearisation (slsqp_body.inc, the block after the QP returns
// mode 4; slsqp_core.c's lsq with n2 = 1): one more variable delta in [0, 1] with the column max(-c, 0) in the
// constraint rows and the weight rho = W.E[n(n+1)/2] (Kraft's l(n3)) as its row of e.  N = n + 1 variables, two more
// bound rows (delta's lower bound follows the other lower bounds, its upper bound the other upper bounds), the rows
// of [g h]; *delta receives the last entry of the step.  What the extra row and column of e
// contribute to the sums of the other variables are products with exact zeros at the end of sums that are never
// -0.0 (delta finite): those terms are left out, every other operation is the serial text's.
//
// unit_e: the caller knows that the factor is diagonal (every entry of l below the diagonal is +0.0: the QP of a solve's
// first iteration and the one after each of Kraft's resets, a seventeenth of all QPs and the longest).  Every sum over
// e's off-diagonal entries then adds products (+-0.0) to a sum that starts +0.0 and stays +0.0, so the forward
// substitution, the walk of the row transform over the blocks before a column's own and the back substitution reduce
// to their last operation, (x - 0.0) / e(i, i), with the bits of the long form.
template <bool AUG = false, class Team>
SPMD_HD int lsq(const Team &T, const Work &W, int H, const double *g, const double *cvals, const double *nz, const double *lo,
                const double *hi, double *s, double *y, double *delta = nullptr, bool unit_e = false)
{
    static_assert(!AUG || MPC_QP_ROWS == 1, "the augmented transform is written for one row per lane");
    const int n = 6 * H, m = 12 * (H + 1), mg = m + 2 * n;
    const int N = n + (AUG ? 1 : 0), ld = mg + (AUG ? 2 : 0);           // ld rows of [g h], columns ld apart
#define EO(r) ((r) * n - ((r) * ((r) - 1)) / 2)
    QP_MARK_START(T);
    // x = e^-1 x in place (W.xs) by back substitution: row i needs the products e(i, j) x(j), j > i, added in ascending j
    // while x(j) becomes known in the opposite order, so the sum itself is a serial chain.  The whole recurrence runs on
    // the team's first wavefront alone: with the vector in registers where lsq_wave.hpp applies (wave_first), else up to
    // 64 threads form the products and the first adds them up (what one wavefront writes to LDS it reads back in program
    // order, so no workgroup barrier separates the 2n steps; the other wavefronts wait at the one barrier after the loop).
    auto back_substitution = [&](bool wave_first) {
        if (unit_e) {
            PAR_FOR(T, i, 0, n) W.xs[i] = (W.xs[i] - 0.0) / W.E[EO(i)];
            T.sync();
            return;
        }
        bool done = false;
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr (Team::kWave) {
            if (wave_first) {                              // lsq_wave.hpp: the same chain at ~8 cycles per term
                if (const int cl = T.chain_lane(); cl >= 0) {
                    const bool ok = lsqwave::back_substitute(W.E, W.xs, W.acc, n);
                    if (cl == 0) W.isc[14] = ok ? 1 : 0;
                }
                T.sync();
                done = W.isc[14] != 0;                     // false: xs is as it was, the team code below runs
                T.sync();
            }
        }
#endif
        (void)wave_first;
        if (!done) {
            const int wl = T.size() < 64 ? T.size() : 64;
            if (T.tid() < wl) {
                for (int i = n - 1; i >= 0; --i) {
                    const double *er = W.E + EO(i) - i;        // er[j] = e(i, j)
                    for (int j = i + 1 + T.tid(); j < n; j += wl) W.piv[j] = er[j] * W.xs[j];
                    T.wave_sync();
                    if (T.tid() == 0) {
                        const double t = sum_in_order(0.0, W.piv + i + 1, n - i - 1);
                        W.xs[i] = (W.xs[i] - t) / er[i];
                    }
                    T.wave_sync();
                }
            }
            T.sync();
        }
    };
    // ---- e = sqrt(d) l^T in place (row r of e is column r of l scaled by sqrt(d_r))
    PAR_FOR(T, i, 0, n) {
        W.piv[i] = sqrt(W.E[EO(i)]);
        W.xs[i] = g[i];
        W.acc[i] = 0.0;
    }
    if (T.tid() == 0) W.isc[8] = 1;
    T.sync();
    // (two indices per row, one for the odd and one for the even columns, eight entries fetched before the first is
    //  written: a row is up to n entries of LDS, and one thread that reads, scales and writes them one by one pays the
    //  LDS round trip per entry)
    PAR_FOR(T, w, 0, 2 * n) {
        const int r = w >> 1, len = n - r;
        double *er = W.E + EO(r);
        const double d = W.piv[r];
        if ((w & 1) == 0) er[0] = d;
        int c = 1 + (w & 1);
        for (; c + 14 < len; c += 16) {
            double v[8];
            SPMD_UNROLL
            for (int q = 0; q < 8; ++q) v[q] = er[c + 2 * q];
            SPMD_UNROLL
            for (int q = 0; q < 8; ++q) er[c + 2 * q] = d * v[q];
        }
        for (; c < len; c += 2) er[c] = d * er[c];
    }
    T.sync();
    QP_MARK(T, 0);   // e in place
    // ---- f: solve e^T f = g by forward substitution, column-oriented: once f(k) is known every later row adds its
    //      term e(k, i) f(k), which it does in the order k = 0, 1, ... of the serial dot product; the thread that
    //      owns row k + 1 (always thread 0) finishes f(k + 1) in the same sweep.  Then negate.
    bool in_registers = false;
    if (unit_e) {
        PAR_FOR(T, i, 0, n) W.ff[i] = (W.xs[i] - 0.0) / W.E[EO(i)];
        T.sync();
        in_registers = true;
    }
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (Team::kWave) {
        if (!unit_e && n <= lsqwave::kMaxN) {               // the team's first wavefront, sums in registers (lsq_wave.hpp)
            if (const int cl = T.chain_lane(); cl >= 0) {
                const bool ok = lsqwave::forward_substitute(W.E, W.xs, W.ff, W.acc, n);
                for (int i = cl; i < n; i += 64) W.acc[i] = 0.0;           // it was scratch there; the team code adds into it
                if (cl == 0) W.isc[14] = ok ? 1 : 0;
            }
            T.sync();
            in_registers = W.isc[14] != 0;                  // false: an operand out of the routine's range, ff not valid
            T.sync();
        }
    }
#endif
    if (!in_registers) {
        if (T.tid() == 0) W.ff[0] = (W.xs[0] - 0.0) / W.E[0];
        T.sync();
        for (int k = 0; k + 1 < n; ++k) {
            const double fk = W.ff[k];
            const double *ek = W.E + EO(k) - k;            // ek[i] = e(k, i)
            PAR_FOR(T, i, k + 1, n) {
                const double t = W.acc[i] + ek[i] * fk;
                W.acc[i] = t;
                if (i == k + 1) W.ff[i] = (W.xs[i] - t) / W.E[EO(i)];
            }
            T.sync();
        }
    }
    PAR_FOR(T, i, 0, n) W.ff[i] = -1.0 * W.ff[i];
    T.sync();
    QP_MARK(T, 1);   // forward substitution
    // ---- LSI: "QR" of the triangular e (pivot arithmetic only, see slsqp_core.c) applied to f: a row touches
    //      nothing but itself, one row per index
    //      (first the rows' scalars, pivots and f, one row per index; then the rows' entries, two indices per row as
    //      above, with the scalars from W.acc, which nothing else uses between the forward substitution and NNLS)
    // (the last row has nothing to its right and is left alone, unless the augmented problem's row follows it)
    PAR_FOR(T, i, 0, N - 1) {
        double *er = W.E + EO(i);
        const double eii = er[0];
        double cl = fabs(eii), up = 0.0, bq = 0.0;
        if (cl > 0.0) {
            const double clinv = 1.0 / cl;
            const double sm = (eii * clinv) * (eii * clinv);
            cl = cl * sqrt(sm);
            if (eii > 0.0) cl = -cl;
            up = eii - cl;
            er[0] = cl;
            bq = up * cl;
            bq = bq < 0.0 ? 1.0 / bq : 0.0;               // 0: the row stays as it is
        }
        W.acc[i] = up;
        W.acc[n + i] = bq;
        if (bq != 0.0) {
            double sm = W.ff[i] * up;
            if (sm != 0.0) {
                sm = sm * bq;
                W.ff[i] = W.ff[i] + sm * up;
            }
        }
    }
    T.sync();
    PAR_FOR(T, w, 0, 2 * (n - 1)) {
        const int i = w >> 1, len = n - i;
        double *er = W.E + EO(i);
        const double up = W.acc[i], bq = W.acc[n + i];
        if (bq == 0.0) continue;
        int c = 1 + (w & 1);
        for (; c + 14 < len; c += 16) {
            double v[8];
            SPMD_UNROLL
            for (int q = 0; q < 8; ++q) v[q] = er[c + 2 * q];
            SPMD_UNROLL
            for (int q = 0; q < 8; ++q) {
                double sm = v[q] * up;
                if (sm != 0.0) {
                    sm = sm * bq;
                    er[c + 2 * q] = v[q] + sm * up;
                }
            }
        }
        for (; c < len; c += 2) {
            double sm = er[c] * up;
            if (sm == 0.0) continue;
            sm = sm * bq;
            er[c] = er[c] + sm * up;
        }
    }
    T.sync();
    PAR_FOR(T, j, 0, N)
        if (!(fabs(W.E[EO(j)]) >= 2.22e-16)) W.isc[8] = 0;       // (EO(n) is where rho sits)
    T.sync();
    if (!W.isc[8]) return 5;
    QP_MARK(T, 2);   // QR rows
#if MPC_QP_FREE_STEP
    // ---- The unconstrained step first.  Four QPs in five end with an empty final set: no row of [g h] has a positive
    //      h, NNLS leaves at its first test, and the answer is x = e^-1 f with every multiplier +0.0 - nothing of the
    //      transformed matrix reaches the result except the signs of h, and h(j) <= 0 says that x satisfies row j of the
    //      constraints.  So that x is computed first (the back substitution the long form would end with, on the same
    //      operands: 0.0 + 1.0 f), and the rows are tested on it directly, G(j) x - h(j) against a margin of
    //      kFreeStepMargin times the magnitude of the terms: where every row passes, the long form's h (the same quantity,
    //      rounded along another path: the two differ by about cond(e) 1e-16 of that magnitude, and the diagonal's spread
    //      is bounded here as well) is negative in every row too, and its result is this one bit for bit.  Where a row
    //      does not pass clearly, the long form decides.
    if constexpr (!AUG) {
        PAR_FOR(T, i, 0, n) W.xs[i] = 0.0 + 1.0 * W.ff[i];
        if (T.tid() == 0) W.isc[4] = 1;
        T.sync();
        back_substitution(in_registers);
        const double a0 = fabs(W.E[0]);
        PAR_FOR(T, i, 0, m + n) {
            bool ok;
            if (i < m) {
                const RowDesc d = describe_row(H, i, cvals, lo, hi);
                double v = 0.0, a = 0.0;
                for (int jj = d.first; jj <= d.last; ++jj) {
                    const double gv = d.kind == 1 ? nz[6 * H + 6 * (jj * H - (jj * (jj - 1)) / 2 - jj - 1 + d.kk) + d.a]
                                                  : nz[6 * jj + d.a];
                    const double t = gv * W.xs[6 * jj + d.a];
                    v = v + t;
                    a = a + fabs(t);
                }
                const double slack = v - d.hval;
                ok = slack > 0.0 && slack >= kFreeStepMargin * (a + fabs(d.hval));
            } else {
                const int q = i - m;
                const double x = W.xs[q], l = lo[q], u = hi[q], e = fabs(W.E[EO(q)]);
                ok = x - l > 0.0 && x - l >= kFreeStepMargin * (fabs(x) + fabs(l)) && u - x > 0.0 &&
                     u - x >= kFreeStepMargin * (fabs(x) + fabs(u)) && e >= kFreeStepSpread * a0 && e * kFreeStepSpread <= a0;
            }
            if (!ok) W.isc[4] = 0;
        }
        T.sync();
#if MPC_QP_FREE_STEP == 2          // development: the long form runs all the same, and says whether the prediction held
        if (T.tid() == 0) { W.isc[6] = W.isc[4]; W.isc[5] = 0; }
        T.sync();
        if (false) {
#else
        if (W.isc[4]) {
#endif
            PAR_FOR(T, i, 0, mg) y[i] = 0.0 + 1.0 * 0.0;
            PAR_FOR(T, i, 0, n) {
                double v = W.xs[i];
                if (v < lo[i]) v = lo[i];
                else if (v > hi[i]) v = hi[i];
                s[i] = v;
            }
            if (T.tid() == 0) { W.isc[3] = 0; W.isc[5] = 1; }
            T.sync();
            QP_MARK(T, 7);   // (the free step's time goes to back substitution + box)
            return 1;
        }
        if (T.tid() == 0) W.isc[5] = 0;
    }
#endif
    // ---- [g h] <- [g e^-1, h - g f], MPC_QP_ROWS rows per index (they share the reads of e), into the matrix NNLS
    //      works on.  Whole wavefronts walk the loop together (a lane beyond the last row repeats a row without
    //      storing), so that on the device a wavefront loads the entries of e once for all its lanes; and the rows are
    //      dealt out in the order of work_order: a row's work begins at its first nonzero block, and a wavefront walks
    //      from the earliest block any of its rows needs, so rows that begin late share wavefronts
    //      (m + n = 18 H + 12 computed rows, a multiple of 2, 3 and 6; the n upper-bound rows are written as twins).
    //      One row per lane and two passes (round 2, late): with two rows per lane the first two wavefronts held every
    //      row that begins at block 0 and walked all of e while the other two had 42 % and 9 % of that to do; with one
    //      row per lane the heavy rows spread over three wavefronts' first pass at 3 instead of 5 instructions per
    //      entry of e, and the second pass is dealt out in the opposite direction, so a wavefront's two walks add up
    //      to 412 .. 481 block-squares instead of 400 .. 9 at the two-row rate: -7 % on whole solves.
    {
        const int per = (mg - n) / MPC_QP_ROWS, width = T.size() < 64 ? T.size() : 64;   // 18H + 12 computed rows
        const int waves = T.size() / width;
#if MPC_QP_SNAKE
        // Groups of `width` rows, in the order of work_order (most work first), each to the wavefront with the least
        // work so far (a group costs the square of the blocks its FIRST row still has to walk; every wavefront runs
        // the same little computation and keeps its own groups, in order).
        const int wv = (T.tid() / width + T.turn()) % waves;
        int load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int base = 0; base < per; base += width) {
            const int p0 = base * MPC_QP_ROWS;                     // the group's first row, in work order
            const int f0 = p0 < n ? 0 : (p0 - n < 12 * H ? (p0 - n) / 12 : H);
            int owner = 0;
            for (int v = 1; v < waves && v < 8; ++v) owner = load[v] < load[owner] ? v : owner;
            load[owner] += (H - f0) * (H - f0) + 1;
            if (owner != wv) continue;
#else
        for (int base = ((T.tid() / width + T.turn()) % waves) * width; base < per; base += T.size()) {
#endif
            const int want = base + T.tid() % width;
            const bool mine = want < per;
            const int i = mine ? want : per - 1;
            const int wide = per - base < width ? per - base : width;        // indices of this wavefront
            int rows[MPC_QP_ROWS];
            double *dst[MPC_QP_ROWS];
            SPMD_UNROLL
            for (int r = 0; r < MPC_QP_ROWS; ++r) {
                rows[r] = work_order(H, base * MPC_QP_ROWS + r * wide + (i - base));
                dst[r] = W.AT + rows[r];
            }
#if defined(__HIP_DEVICE_COMPILE__)
            if constexpr (Team::kWave && MPC_QP_ROWS == 2) {
                typedef __attribute__((address_space(3))) double LdsDouble;
                unsigned e_lds = 0, ff_lds = 0;
                if constexpr (Team::kLdsPool) {
                    e_lds = (unsigned)(unsigned long long)(LdsDouble *)W.E;
                    ff_lds = (unsigned)(unsigned long long)(LdsDouble *)W.ff;
                }
                transform_rows_call<MPC_QP_ROWS, Team::kLdsPool>(W.E, W.ff, e_lds, ff_lds, W.AT, H, cvals, nz, lo, hi, rows[0],
                                                                 rows[MPC_QP_ROWS - 1], mine ? 1 : 0);
            } else
#endif
                transform_rows<MPC_QP_ROWS, true, AUG>(W, H, rows, cvals, nz, lo, hi, dst, mine, W.AT, true, unit_e);
        }
    }
    if constexpr (AUG) {
        PAR_FOR(T, q, 0, 2) delta_row(W, n, ld, q == 1, W.AT + (q == 0 ? m + n : m + n + N));
    }
    T.sync();
    // the upper-bound rows: twins of the lower-bound rows (eight loads in flight per thread: the matrix is in global memory)
    for (int e0 = T.tid(); e0 < n * N; e0 += 8 * T.size()) {
        double v[8];
        SPMD_UNROLL
        for (int q = 0; q < 8; ++q) {
            const int e = e0 + q * T.size(), j = e / n;
            v[q] = e < n * N ? W.AT[(size_t)j * ld + m + (e - j * n)] : 0.0;
        }
        SPMD_UNROLL
        for (int q = 0; q < 8; ++q) {
            const int e = e0 + q * T.size(), j = e / n;
            if (e < n * N) W.AT[(size_t)j * ld + m + N + (e - j * n)] = twin_of(v[q]);
        }
    }
    QP_MARK(T, 3);   // transform
    // ---- LDP: NNLS on the copy of [g h]^T with right-hand side e_{N+1}
    PAR_FOR(T, i, 0, N) W.bb[i] = 0.0;
    if (T.tid() == 0) W.bb[N] = 1.0;
    T.sync();
    double rnorm = 0.0;
    int mode = nnls(T, W, N + 1, ld, ld, &W.sc[2]);
    T.sync();
    rnorm = W.sc[2];
#if MPC_QP_FREE_STEP == 2
    if constexpr (!AUG) {
        if (T.tid() == 0) W.isc[5] = W.isc[6] ? (mode == 1 && W.isc[11] == 0 && W.isc[3] == 0 ? 1 : 2) : 0;   // 2: predicted free, was not
        T.sync();
    }
#endif
    if (mode != 1) return mode;
    if (rnorm <= 0.0) return 4;
    QP_MARK(T, 4);   // NNLS
    // ---- the rows of the transformed [g h] that are in the final set (NNLS overwrote its copy): computed again
    PAR_FOR(T, i, 0, ld)
        if (W.xn[i] != 0.0) {
            if (AUG && (i == m + n || i == ld - 1)) delta_row(W, n, ld, i == ld - 1, W.G + i);
            else {
                const int rows[1] = {i};
                double *const dst[1] = {W.G + i};
                transform_rows<1, false, AUG>(W, H, rows, cvals, nz, lo, hi, dst, true, nullptr, false, unit_e);
            }
        }
    T.sync();
    QP_MARK(T, 5);   // final-set rows
    // ---- column sums against the NNLS solution (column n is h): only the rows in the final set contribute, in
    //      ascending order as in the serial sums over all rows (every other entry of the solution is an exact zero)
    const int nset = W.isc[11];
    if (T.tid() == 0) {                                    // the set, sorted (it is short: insertion sort)
        for (int q = 0; q < nset; ++q) {
            const int v = W.idx[q] - 1;
            int pos = q;
            while (pos > 0 && W.redi[pos - 1] > v) { W.redi[pos] = W.redi[pos - 1]; --pos; }
            W.redi[pos] = v;
        }
    }
    T.sync();
    PAR_FOR(T, j, 0, N + 1) {
        const double *gj = W.G + (size_t)j * ld;
        double t = 0.0;
        for (int q = 0; q < nset; ++q) {
            const int i = W.redi[q];
            const double xi = W.xn[i];
            if (xi != 0.0) t = t + gj[i] * xi;             // rows outside the set: not stored, not read
        }
        W.zz[j] = t;
    }
    T.sync();
    if (T.tid() == 0) {
        const double fac = 1.0 - W.zz[N];                 // 1 - ddot(m, h, y)
        volatile double d = 1.0 + fac;
        W.isc[9] = (d - 1.0 > 0.0) ? 1 : 0;
        W.sc[3] = fac;
    }
    T.sync();
    if (!W.isc[9]) return 4;
    const double fac = 1.0 / W.sc[3];
    PAR_FOR(T, j, 0, N) W.xs[j] = fac * W.zz[j];
    PAR_FOR(T, i, 0, mg) y[i] = 0.0 + fac * W.xn[i];       // (augmented: the bound rows' places are shifted; nobody reads them)
    T.sync();
    QP_MARK(T, 6);   // column sums
    // ---- back to the original variables: x = e^-1 (x + f) by back substitution, then the box.  Row i's sum runs
    //      over j = i + 1, i + 2, ... while x(j) becomes known in the opposite order, so the sum itself is a serial
    //      chain (thread 0); the products are formed by the team first.
    PAR_FOR(T, i, 0, n) W.xs[i] = W.xs[i] + 1.0 * W.ff[i];
    if constexpr (AUG) {                                   // delta first: f = -0.0, an empty sum, e(n, n) = rho
        if (T.tid() == 0) {
            const double xd = W.xs[n] + 1.0 * -0.0;
            W.xs[n] = (xd - 0.0) / W.E[EO(n)];
        }
    }
    T.sync();
    back_substitution(in_registers);
    PAR_FOR(T, i, 0, n) {
        double v = W.xs[i];
        if (v < lo[i]) v = lo[i];
        else if (v > hi[i]) v = hi[i];
        s[i] = v;
    }
    if constexpr (AUG) {
        if (T.tid() == 0) {
            double v = W.xs[n];
            if (v < 0.0) v = 0.0;
            else if (v > 1.0) v = 1.0;
            *delta = v;
        }
    }
    T.sync();
    QP_MARK(T, 7);   // back substitution + box
    return 1;
#undef EO
}

// Column i of the structured dc/dx times r, rows in ascending order: the position row of state j + 1, its velocity
// row, then the position rows of states j + 2 .. H (i = 6 j + a).  Loads are issued eight ahead of the ordered sum.
SPMD_HD inline double jac_column_dot(int H, const double *nz, const double *r, int i)
{
    const int j = i / 6, a = i % 6;
    const double *pos = nz + 6 * H + 6 * (j * H - (j * (j - 1)) / 2 - j - 1) + a;       // pos[6 kk] = entry of state kk
    double t = 0.0;
    if (H <= 24) {
        // short horizons: every entry of the column and of r is requested before the first addition (one trip to
        // global memory instead of one per block of eight); entries beyond the column repeat its last one, unused
        double nv[24], rv[24];
        const double p1 = pos[6 * (j + 1)], r1 = r[12 * (j + 1) + a], nzi = nz[i], r2 = r[12 * (j + 1) + 6 + a];
        SPMD_UNROLL
        for (int q = 0; q < 24; ++q) {
            const int kk = j + 2 + q <= H ? j + 2 + q : H;
            nv[q] = pos[6 * kk];
            rv[q] = r[12 * kk + a];
        }
        t = t + p1 * r1;
        t = t + nzi * r2;
        SPMD_UNROLL
        for (int q = 0; q < 24; ++q)
            if (j + 2 + q <= H) t = t + nv[q] * rv[q];
        return t;
    }
    t = t + pos[6 * (j + 1)] * r[12 * (j + 1) + a];
    t = t + nz[i] * r[12 * (j + 1) + 6 + a];
    int kk = j + 2;
    for (; kk + 7 <= H; kk += 8) {
        double nv[8], rv[8];
        SPMD_UNROLL
        for (int q = 0; q < 8; ++q) { nv[q] = pos[6 * (kk + q)]; rv[q] = r[12 * (kk + q) + a]; }
        SPMD_UNROLL
        for (int q = 0; q < 8; ++q) t = t + nv[q] * rv[q];
    }
    for (; kk <= H; ++kk) t = t + pos[6 * kk] * r[12 * kk + a];
    return t;
}

// Rank-one update of the LDL' factorisation a (packed by columns) with z and sigma; w is work (Fletcher &
// Powell's composite-t method as in slsqp_core.c's ldl).  The serial text interleaves three things per column i:
// the pivot v(i) = z(i) after the earlier columns' eliminations, the scalars (t, alpha, beta, gamma) and the update of
// the column's entries.  They separate without changing one rounded value:
//   A  v = L^-1 z by forward substitution, column-oriented (the eliminations only ever read OLD entries of a);
//      for sigma < 0 the same sweep is the serial text's preliminary pass;
//   B  the scalars: quotients and products per index, the running sums t as one chain of additions on thread 0;
//   C  the entries, one ROW per index: row j walks its columns i < j in order, carrying its own z(j) through the same
//      eliminations again, with no barrier in between.
// Scratch: W.acc[0..3n) and W.xn[0..4n) (both free outside the QP).
template <class Team>
SPMD_HD void ldl(const Team &T, const Work &W, int n, double *a, double *z, double sigma, double *w)
{
    if (sigma == 0.0) return;
    double *vv = W.acc, *dnew = W.acc + n, *quot = W.acc + 2 * n;
    double *alp = W.xn, *bet = W.xn + n, *gam = W.xn + 2 * n, *tpv = W.xn + 3 * n;
#define DIAG(i) ((i) * n - ((i) * ((i) - 1)) / 2)
    // ---- A
    bool in_registers = false;
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (Team::kWave) {
        if (n <= lsqwave::kMaxN) {                          // the team's first wavefront, v in registers (lsq_wave.hpp)
            in_registers = true;
            if (T.chain_lane() >= 0) lsqwave::unit_lower_solve(a, z, vv, n);
            T.sync();
        }
    }
#endif
    if (!in_registers) {
        PAR_FOR(T, i, 0, n) vv[i] = z[i];
        T.sync();
        int ij = 1;
        for (int i = 1; i < n; ++i) {
            const double v = vv[i - 1];
            PAR_FOR(T, j, i + 1, n + 1) vv[j - 1] = vv[j - 1] - v * a[ij - 1 + (j - i)];
            T.sync();
            ij = ij + (n - i) + 1;
        }
    }
    QP_MARK(T, 20);   // ldl A
    // ---- B
    double t0 = 1.0 / sigma;
    if (sigma <= 0.0) {
        PAR_FOR(T, i, 0, n) quot[i] = vv[i] * vv[i] / a[DIAG(i)];
        T.sync();
        if (T.tid() == 0) {
            double t = sum_in_order(t0, quot, n);
            if (t >= 0.0) t = 2.22e-16 / sigma;
            int j = n;
            for (; j >= 16; j -= 16) {                      // sixteen terms fetched together, the chain in its order
                double qv[16];
                SPMD_UNROLL
                for (int q = 0; q < 16; ++q) qv[q] = quot[j - 1 - q];
                SPMD_UNROLL
                for (int q = 0; q < 16; ++q) {
                    w[j - 1 - q] = t;
                    t = t - qv[q];
                }
            }
            for (; j >= 1; --j) {
                w[j - 1] = t;
                t = t - quot[j - 1];
            }
            W.sc[4] = t;
        }
        T.sync();
        t0 = W.sc[4];
        PAR_FOR(T, i, 0, n) tpv[i] = w[i];
    } else {
        PAR_FOR(T, i, 0, n) {
            const double delta = vv[i] / a[DIAG(i)];
            quot[i] = delta * vv[i];
        }
        T.sync();
        if (T.tid() == 0) {
            double t = t0;
            int i = 0;
            for (; i + 16 <= n; i += 16) {                  // sixteen terms fetched together, the chain in its order
                double qv[16];
                SPMD_UNROLL
                for (int q = 0; q < 16; ++q) qv[q] = quot[i + q];
                SPMD_UNROLL
                for (int q = 0; q < 16; ++q) {
                    t = t + qv[q];
                    tpv[i + q] = t;
                }
            }
            for (; i < n; ++i) {
                t = t + quot[i];
                tpv[i] = t;
            }
        }
    }
    T.sync();
    PAR_FOR(T, i, 0, n) {
        const double aii = a[DIAG(i)], tp = tpv[i], t = i == 0 ? t0 : tpv[i - 1];
        const double delta = vv[i] / aii;
        const double alpha = tp / t;
        dnew[i] = alpha * aii;
        alp[i] = alpha;
        bet[i] = delta / tp;
        gam[i] = t / tp;
    }
    T.sync();
    QP_MARK(T, 21);   // ldl B
    // ---- C
    PAR_FOR(T, j, 1, n) {
        double zj = z[j];
        for (int i = 0; i < j; ++i) {
            const int p = DIAG(i) + (j - i);
            const double v = vv[i], u = a[p];
            if (alp[i] > 4.0) {
                a[p] = gam[i] * u + bet[i] * zj;
                zj = zj - v * u;
            } else {
                zj = zj - v * u;
                a[p] = u + bet[i] * zj;
            }
        }
    }
    T.sync();
    PAR_FOR(T, i, 0, n) {
        a[DIAG(i)] = dnew[i];
        z[i] = vv[i];
    }
    T.sync();
#undef DIAG
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
        u[i] = g[i] - jac_column_dot(H, nz, r, i) - v[i];
    }
    T.sync();
    PAR_FOR(T, i, 0, n) {
        const int off = i * n - (i * (i - 1)) / 2;
        double h1 = 0.0;
        {
            int j = i + 1;
            for (; j + 8 <= n; j += 8) {                    // loads first, sum in order
                double lv[8], sv[8];
                SPMD_UNROLL
                for (int q = 0; q < 8; ++q) { lv[q] = l[off + (j + q - i)]; sv[q] = s[j + q]; }
                SPMD_UNROLL
                for (int q = 0; q < 8; ++q) h1 = h1 + lv[q] * sv[q];
            }
            for (; j < n; ++j) h1 = h1 + l[off + (j - i)] * s[j];
        }
        double vi = s[i] + h1;
        vi = l[off] * vi;
        vold[i] = vi;
    }
    T.sync();
    PAR_FOR(T, i, 0, n) {
        double h1 = 0.0;
        {
            int j = 0;
            for (; j + 8 <= i; j += 8) {
                double lv[8], vv8[8];
                SPMD_UNROLL
                for (int q = 0; q < 8; ++q) {
                    const int jq = j + q;
                    lv[q] = l[jq * n - (jq * (jq - 1)) / 2 + (i - jq)];
                    vv8[q] = vold[jq];
                }
                SPMD_UNROLL
                for (int q = 0; q < 8; ++q) h1 = h1 + lv[q] * vv8[q];
            }
            for (; j < i; ++j) h1 = h1 + l[j * n - (j * (j - 1)) / 2 + (i - j)] * vold[j];
        }
        v[i] = vold[i] + h1;
    }
    T.sync();
    bool summed = false;
    double h1w = 0.0, h2w = 0.0;
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (Team::kWave) {
        if (n <= lsqwave::kMaxN) {                          // the two ordered sums as DPP chains on the first wavefront
            summed = true;
            if (T.tid() < 64) {
                const int c16 = lsqwave::lane_here() & 15, pad = lsqwave::kMaxN - n;
                double P1[8], P2[8];
                SPMD_UNROLL
                for (int rr = 0; rr < 8; ++rr) {
                    const int i = 16 * rr + c16 - pad;
                    const double si = i >= 0 ? s[i] : 0.0;
                    P1[rr] = i >= 0 ? si * u[i] : 0.0;
                    P2[rr] = i >= 0 ? si * v[i] : 0.0;
                }
                h1w = lsqwave::ordered_sum_from(0.0, P1, pad);
                h2w = lsqwave::ordered_sum_from(0.0, P2, pad);
            }
        }
    }
#endif
    if (T.tid() == 0) {
        double h1 = h1w, h2 = h2w;
        if (!summed) {
            for (int i = 0; i < n; ++i) h1 = h1 + s[i] * u[i];
            for (int i = 0; i < n; ++i) h2 = h2 + s[i] * v[i];
        }
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
    QP_MARK(T, 12);   // BFGS products and damping
    ldl(T, W, n, l, u, +1.0 / h1, v);
    QP_MARK(T, 13);   // first rank-one update
    ldl(T, W, n, l, v, -1.0 / h2, u);
    QP_MARK(T, 14);   // second rank-one update
}

}  // namespace lsqspmd
