// MPC_ALGO_PREFIX — one workgroup per scenario (several scenarios when the horizon is short).
//
// scipy's forward difference perturbs ONE control u_j[a] per rollout, and a perturbation at step j
// leaves every state before step j untouched.  So the unperturbed rollout is evaluated once
// (phase 1: one lane per horizon step, per-step results parked in LDS), and every perturbed
// rollout restarts from the shared prefix at its own step j (phase 2: one lane per PAIR of
// perturbations (j, H-1-j) of one joint, so that every lane walks H+3 steps in all and the wave
// stays full).  Per step a perturbed lane needs one new sin/cos (its own joint); the other five
// joints, the platform yaw and Jc u_k come from LDS.  Phase 3 writes the dense dc/dx (mostly
// structural zeros) cooperatively with 16-byte coalesced stores from the non-zeros staged in LDS.
//
// Arithmetic: fp64, explicit fma() in the kinematic chain (this unit is compiled with
// -ffp-contract=off, so phase 1 and phase 2 run bit-identical code on identical inputs and a zero
// step reproduces f0 exactly); sums that the reference orders (stage terms, stage totals, Euler
// updates) keep the reference's order.  Differences from the CPU oracle are a few ulp of f.
//
// Restates: MPC:665-692 (_cost_function), MPC:249-381 (stage/terminal cost), DYN:101-128 (Euler
// step), MPC:388-411 (constraint), NUMDIFF:44-72,498-524,584-625 (scipy's forward difference).
#include "mpc_device.hpp"
#include "mpc_kernels.hpp"

namespace {

using namespace mpcdev;

constexpr int kMaxLdsBytes = 64 * 1024;    // fused mode: keeps >= 2 workgroups per CU
constexpr int kMaxLdsLarge = 160 * 1024;   // compact mode: one workgroup may take the whole CU

struct PrefixArgs {
    const double *decay;
    mpc_scenarios sc;
    const double *x;
    double *cost;
    double *grad;
    double *con;
    double *jac;
    double *compact;  // MODE 2: per scenario [6H velocity quotients][3H(H+1) position quotients]
    int cstride;      // doubles per scenario in `compact`
    int tmajor;       // MODE 2, internal workspace only: quotients laid out [iteration][phase-2 lane] (coalesced stores)
    int S;            // scenarios per workgroup
    int items;        // phase-2 lanes per scenario = 6 * ceil(H/2)
    // ROWS form (device-resident solves): evaluate rows list[0 .. *count_dev) of every array in place
    const int32_t *list;
    const int32_t *count_dev;
    int single_row = -1;  // ROWS form without a list (kernels_team.hip): this one row
};

// How the forward-difference work is organised:
//   kEval    f and c only (phase 1)
//   kFused   FD, the non-zeros of dc/dx staged in LDS, dense dc/dx written by the same workgroup
//   kCompact FD for long horizons: the non-zeros go to a global workspace, expand_kernel writes dc/dx
enum Mode { kEval = 0, kFused = 1, kCompact = 2 };

// Per-step records that the perturbed rollouts read with a different step in every group of six lanes are 14 doubles
// apart, not 12: with 96-byte records steps k and k + 8 start in the same LDS bank (2-way conflicts on a fifth of
// the reads), with 112-byte records ten consecutive steps start in ten different banks four apart.
constexpr int kBtStride = 14;    // platform trajectory: [P(3), cos yaw, sin yaw, yaw, V(3), Omega(3), pad(2)]
constexpr int kScsStride = 14;   // sin[6], cos[6] of the joint angles, pad(2)

// LDS layout of one scenario (offsets in doubles)
struct Layout {
    int xs, bt, ys, st0, tgt, Jc, up, scs, po, qs, qn, Cp, vv, tb, lb, jvel, jpos, total;
};

__host__ __device__ inline Layout make_layout(int H, int mode, bool same_dt)
{
    Layout Y;
    int o = 0;
    Y.xs = o;   o += 6 * H;         // control sequence
    Y.bt = o;   o += kBtStride * H; // platform trajectory; cos / sin of the yaw replace roll and pitch, which nothing reads
    Y.ys = Y.bt + 3;                // cos, sin of the platform yaw of step k at ys + kBtStride k
    Y.st0 = o;  o += 18;
    Y.tgt = o;  o += 6;
    Y.Jc = o;   o += 36;
    Y.up = o;   o += 6;
    Y.scs = o;  o += kScsStride * (H + 1);  // sin[6], cos[6] of the unperturbed joint angles per step
    Y.po = o;   o += 6 * (H + 1);   // unperturbed end-effector pose per step
    Y.qs = o;   o += 6 * (H + 1);   // unperturbed joint angles, cost rollout (dt_cost)
    Y.qn = o;   o += same_dt ? 0 : 6 * (H + 1);  // same, constraint rollout (dt_con)
    Y.Cp = o;   o += H + 2;         // Cp[k] = L_0 + ... + L_{k-1}; Cp[H+1] = f0
    Y.vv = o;   o += 6 * H;         // Jc @ u_k
    Y.tb = o;                       // phase-1 scratch: Euler increments [6H] + step costs [H+1]
    Y.lb = o + 6 * H;
    const int scratch = 7 * H + 1;
    if (mode == kFused) {
        o += o & 1;                 // 16-byte alignment of the two arrays phase 3 reads as pairs
        Y.jvel = o; o += 6 * H;     // d(vel row of state j+1)/d u_j[a], [j][a]
        Y.jpos = o;                 // d(pos row of state k)/d u_j[a], k > j, packed [tri(j,k)][a]
        const int tri = 3 * H * (H + 1);
        // phase-1 scratch lives inside jpos when it fits (it is dead before phase 2 starts)
        Y.tb = Y.jpos;
        Y.lb = Y.jpos + 6 * H;
        o += (tri > scratch) ? tri : scratch;
    } else {
        Y.jvel = Y.jpos = o;
        o += scratch;
    }
    o += o & 1;
    Y.total = o;
    return Y;
}

// ---- kinematics specialised to the UR twist pattern ------------------------------------------
// alpha = (+90, 0, 0, +90, -90, 0) deg, a = (0, a2, a3, 0, 0, 0), d = (d1, 0, 0, d4, d5, d6).
// Dropping the terms that are exactly zero / one leaves the generic DH product's values unchanged.
template <int CA, int SA, bool HAS_A, bool HAS_D>
__device__ __forceinline__ void ur_link(double (&R)[3][3], double (&o)[3], double c, double s, double a, double d)
{
    const double tx = a * c, ty = a * s;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const double r0 = R[r][0], r1 = R[r][1], r2 = R[r][2];
        if (HAS_A) {
            double inc = fma(r1, ty, r0 * tx);
            if (HAS_D) inc = fma(r2, d, inc);
            o[r] = o[r] + inc;
        } else if (HAS_D) {
            o[r] = fma(r2, d, o[r]);
        }
        R[r][0] = fma(r1, s, r0 * c);
        if (CA == 1) {  // alpha = 0
            R[r][1] = fma(r1, c, -(r0 * s));
            R[r][2] = r2;
        } else if (SA > 0) {  // alpha = +90 deg
            R[r][1] = r2;
            R[r][2] = fma(r0, s, -(r1 * c));
        } else {  // alpha = -90 deg
            R[r][1] = -r2;
            R[r][2] = fma(r1, c, -(r0 * s));
        }
    }
}

// w = J(q) @ u accumulated while walking the chain:
//   w_ang = sum_i u_i z_i,  w_lin = sum_i u_i z_i x (pe - o_i) = w_ang x pe - sum_i (u_i z_i) x o_i
__device__ __forceinline__ void ur_twist(const mpc_params &P, const double (&sn)[6], const double (&cs)[6],
                                         const double (&u)[6], double (&w)[6])
{
    double R[3][3] = {{P.base_c, -P.base_s, 0.0}, {P.base_s, P.base_c, 0.0}, {0.0, 0.0, 1.0}};
    double o[3] = {0.0, 0.0, 0.0};
    double om[3] = {0.0, 0.0, 0.0}, mm[3] = {0.0, 0.0, 0.0};
    auto joint = [&](int i) {
        const double z0 = u[i] * R[0][2], z1 = u[i] * R[1][2], z2 = u[i] * R[2][2];
        om[0] += z0; om[1] += z1; om[2] += z2;
        if (i > 0) {  // o_0 = 0
            mm[0] = fma(z1, o[2], fma(-z2, o[1], mm[0]));
            mm[1] = fma(z2, o[0], fma(-z0, o[2], mm[1]));
            mm[2] = fma(z0, o[1], fma(-z1, o[0], mm[2]));
        }
    };
    joint(0); ur_link<0, 1, false, true>(R, o, cs[0], sn[0], 0.0, P.dh_d[0]);
    joint(1); ur_link<1, 0, true, false>(R, o, cs[1], sn[1], P.dh_a[1], 0.0);
    joint(2); ur_link<1, 0, true, false>(R, o, cs[2], sn[2], P.dh_a[2], 0.0);
    joint(3); ur_link<0, 1, false, true>(R, o, cs[3], sn[3], 0.0, P.dh_d[3]);
    joint(4); ur_link<0, -1, false, true>(R, o, cs[4], sn[4], 0.0, P.dh_d[4]);
    joint(5); ur_link<1, 0, false, true>(R, o, cs[5], sn[5], 0.0, P.dh_d[5]);
    double pe[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) pe[r] = fma(R[r][2], P.tool[2], fma(R[r][1], P.tool[1], fma(R[r][0], P.tool[0], o[r])));
    w[0] = fma(om[1], pe[2], -(om[2] * pe[1])) - mm[0];
    w[1] = fma(om[2], pe[0], -(om[0] * pe[2])) - mm[1];
    w[2] = fma(om[0], pe[1], -(om[1] * pe[0])) - mm[2];
    w[3] = om[0]; w[4] = om[1]; w[5] = om[2];
}

// Stage cost L_k (stage = true, MPC:249-328) or terminal cost (MPC:330-381) with one code path:
// the terminal form is the stage form with decay -> (terminal_factor, 1), no control terms, and
// V = J(q_H) dq_H instead of Jc u_k.  Adding the three +0.0 terms leaves the value unchanged.
// Products and sums inside one term are fused (a few ulp from the reference's unfused order);
// the order in which the seven terms are added is the reference's.
__device__ __forceinline__ double sumsq3f(double a, double b, double c) { return fma(c, c, fma(b, b, a * a)); }

__device__ __forceinline__ double unit_cost(const mpc_params &P, bool stage, const double (&V)[6],
                                            const double (&u)[6], const double (&p)[3], const double (&o)[3],
                                            const double *b, double cy, double sy, const double *tgt, double dk,
                                            bool damp, const double *up, bool smooth)
{
    const double r0 = p[0] - b[0], r1 = p[1] - b[1], r2 = p[2] - b[2];
    const double O0 = b[9], O1 = b[10], O2 = b[11];
    // -(V + Omega x r), -(Omega), rotated by R(psi) = [[c,s,0],[-s,c,0],[0,0,1]] (MPC:270-283,714-720)
    const double w0 = -(b[6] + fma(O1, r2, -(O2 * r1)));
    const double w1 = -(b[7] + fma(O2, r0, -(O0 * r2)));
    const double w2 = -(b[8] + fma(O0, r1, -(O1 * r0)));
    const double tl0 = fma(sy, w1, cy * w0), tl1 = fma(cy, w1, -(sy * w0)), tl2 = w2;
    const double ta0 = -fma(sy, O1, cy * O0), ta1 = fma(sy, O0, -(cy * O1)), ta2 = -O2;
    const double fv = stage ? dk : P.terminal_factor;
    const double fp = stage ? dk : 1.0;
    const double c_lin = P.w_linear * fv * sumsq3f(V[0] - tl0, V[1] - tl1, V[2] - tl2);
    const double c_ang = P.w_angular * fv * sumsq3f(V[3] - ta0, V[4] - ta1, V[5] - ta2);
    const double c_pos = P.w_position * fp * sumsq3f(p[0] - tgt[0], p[1] - tgt[1], p[2] - tgt[2]);
    const double c_ori = P.w_orientation * fp * sumsq3f(o[0] - tgt[3], o[1] - tgt[4], o[2] - tgt[5]);
    double uu = u[0] * u[0], dd = (u[0] - up[0]) * (u[0] - up[0]);
#pragma unroll
    for (int j = 1; j < 6; ++j) {
        uu = fma(u[j], u[j], uu);
        const double d = u[j] - up[j];
        dd = fma(d, d, dd);
    }
    const double c_ctl = stage ? P.w_control * uu : 0.0;
    const double c_damp = damp ? P.w_damping * dd : 0.0;
    const double c_smooth = smooth ? P.w_smooth * uu : 0.0;
    return (((((c_lin + c_ang) + c_pos) + c_ori) + c_ctl) + c_damp) + c_smooth;
}

// num / dx from a reciprocal computed once per perturbation, with one residual correction step:
// equal to the correctly rounded quotient except for rare last-place ties.
// sin and cos of a joint or yaw angle.  The library routine spends ~100 VALU instructions per call on
// double-double arithmetic and a large-argument path; angles here are a few radians (|q| <= 2 pi by
// the joint limits), so two-constant Cody-Waite reduction by pi/2 with fma and the classic minimax
// polynomials on [-pi/4, pi/4] (Sun fdlibm's kernel coefficients) give results within an ulp or so of
// the correctly rounded value in about a third of the instructions.  Beyond ~1e5 rad the reduction
// loses accuracy gradually (no large-argument path: that code would cost the kernel a wave of
// occupancy in registers); inf and NaN give NaN like the library.
__device__ __forceinline__ void sincos_small(double x, double *sn, double *cs)
{
    const double k = rint(x * 0.63661977236758134308);            // 2/pi
    double r = fma(-k, 1.57079632679489655800e+00, x);              // pi/2, high part
    r = fma(-k, 6.12323399573676603587e-17, r);                     // pi/2, low part
    const double z = r * r;
    double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = fma(z, ps, 2.75573137070700676789e-06);
    ps = fma(z, ps, -1.98412698298579493134e-04);
    ps = fma(z, ps, 8.33333333332248946124e-03);
    ps = fma(z, ps, -1.66666666666666324348e-01);
    const double s = fma(r * z, ps, r);
    double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = fma(z, pc, -2.75573143513906633035e-07);
    pc = fma(z, pc, 2.48015872894767294178e-05);
    pc = fma(z, pc, -1.38888888888741095749e-03);
    pc = fma(z, pc, 4.16666666666666019037e-02);
    const double hz = 0.5 * z, a = 1.0 - hz;
    const double c = a + (((1.0 - a) - hz) + (z * z) * pc);
    const int q = (int)k & 3;
    const double s1 = (q & 1) ? c : s, c1 = (q & 1) ? s : c;
    *sn = (q & 2) ? -s1 : s1;
    *cs = ((q + 1) & 2) ? -c1 : c1;
}

__device__ __forceinline__ double quot(double num, double dx, double rdx)
{
    const double q = num * rdx;
    return fma(fma(-dx, q, num), rdx, q);
}

__device__ __forceinline__ double pick6(const double (&a)[6], int j)
{
    double v = a[0];
#pragma unroll
    for (int c = 1; c < 6; ++c) v = (j == c) ? a[c] : v;
    return v;
}

// The work of one workgroup (thread tid of bs, scenarios s0 .. s0 + A.S - 1 of the batch or of the request list, lds:
// the workgroup's 16-byte aligned LDS of geometry(P, MODE).lds bytes).  A function of its own so that a kernel which
// does more than evaluate (kernels_team.hip: whole solves per workgroup) can run the very same text; every thread of
// the workgroup must call it (it contains workgroup barriers).
template <int MODE, bool SAME_DT, bool ROWS>
__device__ __forceinline__ void prefix_body(const mpc_params &P, const PrefixArgs &A, double *lds, const int tid, const int bs,
                                            const int s0)
{
    const int H = P.horizon, n = 6 * H, m = 12 * (H + 1), L = H + 1;
    constexpr bool FD = MODE != kEval;
    const Layout Y = make_layout(H, MODE, SAME_DT);
    double *dec = lds;  // [H], shared by the scenarios of the workgroup
    double *scen = lds + ((H + 1) & ~1);
    const bool one = ROWS && A.single_row >= 0;
    const int ns = one ? 1 : min(A.S, (ROWS ? *A.count_dev : A.sc.count) - s0);
    if (ROWS && ns <= 0) return;                 // grid sized by an upper bound of the list length
    // row of local scenario ls in the caller's arrays: consecutive, or taken from the request list
    auto row = [&](int ls) -> size_t {
        return ROWS ? (one ? (size_t)A.single_row : (size_t)A.list[s0 + ls]) : (size_t)(s0 + ls);
    };

    // ---- stage the scenarios' inputs: rows of consecutive scenarios are contiguous in HBM ------
    for (int i = tid; i < H; i += bs) dec[i] = A.decay[i];
    for (int f = tid; f < ns * n; f += bs) {
        const int ls = f / n;
        scen[ls * Y.total + Y.xs + (f - ls * n)] = A.x[ROWS ? row(ls) * n + (f - ls * n) : (size_t)s0 * n + f];
    }
    for (int f = tid; f < ns * 12 * H; f += bs) {
        const int ls = f / (12 * H);
        const int e = f - ls * 12 * H, k = e / 12;
        scen[ls * Y.total + Y.bt + kBtStride * k + (e - 12 * k)] =
            A.sc.base_traj[ROWS ? row(ls) * 12 * H + e : (size_t)s0 * 12 * H + f];
    }
    for (int f = tid; f < ns * 18; f += bs)
        scen[(f / 18) * Y.total + Y.st0 + f % 18] = A.sc.state0[ROWS ? row(f / 18) * 18 + f % 18 : (size_t)s0 * 18 + f];
    for (int f = tid; f < ns * 6; f += bs)
        scen[(f / 6) * Y.total + Y.tgt + f % 6] = A.sc.target[ROWS ? row(f / 6) * 6 + f % 6 : (size_t)s0 * 6 + f];
    for (int f = tid; f < ns * 36; f += bs)
        scen[(f / 36) * Y.total + Y.Jc + f % 36] = A.sc.Jc[ROWS ? row(f / 36) * 36 + f % 36 : (size_t)s0 * 36 + f];
    const bool any_prev = A.sc.u_prev != nullptr && A.sc.has_prev != nullptr;
    for (int f = tid; f < ns * 6; f += bs)
        scen[(f / 6) * Y.total + Y.up + f % 6] =
            any_prev ? A.sc.u_prev[ROWS ? row(f / 6) * 6 + f % 6 : (size_t)s0 * 6 + f] : 0.0;
    __syncthreads();
    for (int f = tid; f < ns * H; f += bs) {
        const int ls = f / H, k = f - ls * H;
        double *S = scen + ls * Y.total;
        double sy, cy;
        sincos_small(S[Y.bt + kBtStride * k + 5], &sy, &cy);
        S[Y.ys + kBtStride * k] = cy;
        S[Y.ys + kBtStride * k + 1] = sy;
    }

    // ================= phase 1: the unperturbed rollout, one lane per step =====================
    const bool act1 = tid < ns * L;
    const int ls1 = act1 ? tid / L : 0;
    const int k1 = act1 ? tid - ls1 * L : 0;
    double *S1 = scen + ls1 * Y.total;
    const bool has_prev1 = any_prev && A.sc.has_prev[row(ls1)] != 0;
    double u1[6], w1[6], vc1[6];
    if (act1) {
        const int k = k1;
        double qc[6], qn[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) { qc[j] = S1[Y.st0 + j]; qn[j] = qc[j]; }
        for (int jj = 0; jj < k; ++jj) {
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const double uj = S1[Y.xs + 6 * jj + j];
                qc[j] = qc[j] + uj * P.dt_cost;            // DYN:105
                if (!SAME_DT) qn[j] = qn[j] + uj * P.dt_con;
            }
        }
        double sn[6], cs[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            sincos_small(qc[j], &sn[j], &cs[j]);
            S1[Y.qs + 6 * k + j] = qc[j];
            if (!SAME_DT) S1[Y.qn + 6 * k + j] = qn[j];
            S1[Y.scs + kScsStride * k + j] = sn[j];
            S1[Y.scs + kScsStride * k + 6 + j] = cs[j];
        }
        const int ku = (k < H) ? k : H - 1;
#pragma unroll
        for (int j = 0; j < 6; ++j) u1[j] = S1[Y.xs + 6 * ku + j];
        ur_twist(P, sn, cs, u1, w1);
        if (k < H) {
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                vc1[c] = dot6(S1 + Y.Jc + 6 * c, u1);     // MPC:260-261
                S1[Y.vv + 6 * k + c] = vc1[c];
                S1[Y.tb + 6 * k + c] = w1[c] * P.dt_cost;  // DYN:119-120
            }
        } else {
#pragma unroll
            for (int c = 0; c < 6; ++c) vc1[c] = w1[c];    // terminal: J(q_H) dq_H (MPC:344-345)
        }
        if (A.con != nullptr) {                            // MPC:394-409
            double2 *dst = reinterpret_cast<double2 *>(A.con + row(ls1) * m + 12 * k);
            double c0[12];
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const double dq = (k == 0) ? S1[Y.st0 + 6 + j] : S1[Y.xs + 6 * (k - 1) + j];
                c0[j] = con_pos(P, j, SAME_DT ? qc[j] : qn[j]);
                c0[6 + j] = con_vel(P, j, dq);
            }
#pragma unroll
            for (int t = 0; t < 6; ++t) dst[t] = make_double2(c0[2 * t], c0[2 * t + 1]);
        }
    }
    __syncthreads();
    if (act1) {
        const int k = k1;
        double p[3] = {S1[Y.st0 + 12], S1[Y.st0 + 13], S1[Y.st0 + 14]};
        double o[3] = {S1[Y.st0 + 15], S1[Y.st0 + 16], S1[Y.st0 + 17]};
        for (int jj = 0; jj < k; ++jj) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                p[c] = p[c] + S1[Y.tb + 6 * jj + c];
                o[c] = o[c] + S1[Y.tb + 6 * jj + 3 + c];
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) { S1[Y.po + 6 * k + c] = p[c]; S1[Y.po + 6 * k + 3 + c] = o[c]; }
        const int kb = (k < H) ? k : H - 1;                // terminal uses B_{H-1} (MPC:689)
        const double Lk = unit_cost(P, k < H, vc1, u1, p, o, S1 + Y.bt + kBtStride * kb, S1[Y.ys + kBtStride * kb],
                                    S1[Y.ys + kBtStride * kb + 1], S1 + Y.tgt, dec[kb], k > 0 && k < H && has_prev1,
                                    S1 + Y.up, k > 0 && k < H && H > 1);
        S1[Y.lb + k] = Lk;
    }
    __syncthreads();
    if (act1) {
        const int k = k1;
        double tot = 0.0;
        for (int jj = 0; jj < k; ++jj) tot = tot + S1[Y.lb + jj];  // MPC:683, left to right
        if (k == H) {
            const double f0 = tot + S1[Y.lb + H];                   // MPC:690
            A.cost[row(ls1)] = f0;
            S1[Y.Cp + H + 1] = f0;
        }
        // Cp lives apart from lb (which aliases jpos): every lane has read lb before any writes jpos
        S1[Y.Cp + k] = tot;
    }
    if (!FD) return;
    __syncthreads();

    // ================= phase 2: perturbed rollouts from the shared prefix ======================
    const bool act2 = tid < ns * A.items;
    const int ls2 = act2 ? tid / A.items : 0;
    const int it = act2 ? tid - ls2 * A.items : 0;
    const int pi = it / 6, a = it - 6 * pi;
    double *S2 = scen + ls2 * Y.total;
    const bool has_prev2 = any_prev && A.sc.has_prev[row(ls2)] != 0;
    const int j_first = pi, j_second = H - 1 - pi;
    const int npert = (j_second > j_first) ? 2 : 1;
    const double lim = pick6(P.v_lim, a), qlo = pick6(P.q_lo, a), qhi = pick6(P.q_hi, a);
    const double f0 = S2[Y.Cp + H + 1];

    bool active = act2, need_init = true;
    int pert = 0, j = 0, k = 0, var = 0, tri_j = 0;
    double acc = 0.0, xv = 0.0, x0v = 0.0, dx = 1.0, rdx = 1.0;
    double p[3] = {0, 0, 0}, o[3] = {0, 0, 0};
    double qc = 0.0, qnp = 0.0, qnb = 0.0;
    const int iters = H + 3;
    for (int step = 0; step < iters; ++step) {
        if (!active) continue;
        if (need_init) {
            j = (pert == 0) ? j_first : j_second;
            var = 6 * j + a;
            x0v = S2[Y.xs + var];
            xv = x0v + fd_step(x0v, -lim, lim, P.fd_eps);   // NUMDIFF:594
            dx = xv - x0v;                                   // NUMDIFF:596
            rdx = 1.0 / dx;
            k = j;
            acc = S2[Y.Cp + j];
#pragma unroll
            for (int c = 0; c < 3; ++c) { p[c] = S2[Y.po + 6 * j + c]; o[c] = S2[Y.po + 6 * j + 3 + c]; }
            qc = S2[Y.qs + 6 * j + a];
            qnb = SAME_DT ? qc : S2[Y.qn + 6 * j + a];
            qnp = qnb;
            tri_j = j * H - (j * (j - 1)) / 2 - j - 1;  // + k gives the packed index of (j, k)
            need_init = false;
        }
        // ---- step k of this perturbed rollout ----
        const int ku = (k < H) ? k : H - 1;
        double u[6], sn[6], cs[6];
        double sa, ca;
        sincos_small(qc, &sa, &ca);
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            const double ub = S2[Y.xs + 6 * ku + c];
            u[c] = (c == a && ku == j) ? xv : ub;
            const double sb = S2[Y.scs + kScsStride * k + c], cb = S2[Y.scs + kScsStride * k + 6 + c];
            sn[c] = (c == a) ? sa : sb;
            cs[c] = (c == a) ? ca : cb;
        }
        double w[6], V[6];
        ur_twist(P, sn, cs, u, w);
        const double du = (k == j) ? dx : 0.0;
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            // Jc @ u'_k = Jc @ u_k + Jc[:,a] (u'_k[a] - u_k[a]); the correction is zero off step j
            const double vs = fma(S2[Y.Jc + 6 * c + a], du, S2[Y.vv + 6 * ku + c]);
            V[c] = (k < H) ? vs : w[c];
        }
        const double Lk = unit_cost(P, k < H, V, u, p, o, S2 + Y.bt + kBtStride * ku, S2[Y.ys + kBtStride * ku],
                                    S2[Y.ys + kBtStride * ku + 1], S2 + Y.tgt, dec[ku], k > 0 && k < H && has_prev2,
                                    S2 + Y.up, k > 0 && k < H && H > 1);
        acc = acc + Lk;
        if (k < H) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                p[c] = p[c] + w[c] * P.dt_cost;
                o[c] = o[c] + w[3 + c] * P.dt_cost;
            }
            const double ub = S2[Y.xs + 6 * k + a];
            const double ua = (k == j) ? xv : ub;
            qc = qc + ua * P.dt_cost;
            if (SAME_DT) {
                qnp = qc;
                qnb = S2[Y.qs + 6 * (k + 1) + a];
            } else {
                qnp = qnp + ua * P.dt_con;
                qnb = qnb + ub * P.dt_con;
            }
            // column `var` of dc/dx, rows of state k+1 (the other rows of the column are +-0)
            const double c1 = fmin(qnp - qlo, qhi - qnp), c0 = fmin(qnb - qlo, qhi - qnb);
            const double jp = quot(c1 - c0, dx, rdx);
            const double jv = quot((lim - fabs(xv)) - (lim - fabs(x0v)), dx, rdx);
            if (MODE == kFused) {
                S2[Y.jpos + 6 * (tri_j + k + 1) + a] = jp;
                if (k == j) S2[Y.jvel + var] = jv;
            } else if (A.jac != nullptr) {
                double *cz = A.compact + row(ls2) * A.cstride;
                if (A.tmajor) {
                    cz[step * A.items + it] = jp;   // one contiguous run per wavefront and iteration
                } else {
                    cz[6 * H + 6 * (tri_j + k + 1) + a] = jp;
                    if (k == j) cz[var] = jv;
                }
            }
            k = k + 1;
        } else {
            if (MODE == kCompact && A.jac != nullptr && A.tmajor)   // the terminal iteration's slot carries the velocity quotient
                A.compact[row(ls2) * A.cstride + step * A.items + it] =
                    quot((lim - fabs(xv)) - (lim - fabs(x0v)), dx, rdx);
            if (A.grad != nullptr) A.grad[row(ls2) * n + var] = quot(acc - f0, dx, rdx);  // NUMDIFF:619
            pert = pert + 1;
            if (pert < npert) need_init = true; else active = false;
        }
    }
    if (MODE != kFused || A.jac == nullptr) return;
    __syncthreads();

    // ================= phase 3: dense dc/dx, column-major m x n per scenario ====================
    // Each wave takes scenarios round-robin; per pass a lane owns two consecutive rows (one
    // 16-byte store) of EVERY column: rows (t, t+1) of state kk, i.e. the position (t < 6) or
    // velocity (t >= 6) rows of the joint pair (2gp, 2gp+1).  Column (jj, aa) is non-zero in this
    // piece only for aa in that pair, and only if kk > jj (position) or kk == jj+1 (velocity), so
    // one 16-byte LDS read per (jj, pass) feeds the lane's six stores of that step.
    // Structural zeros are written as +0.0 (scipy's quotient (c - c)/dx gives -0.0 where the step
    // was flipped; the two compare equal).
    // Columns are written in ascending order with all of a column's rows back to back, so that the
    // 128-byte lines shared by neighbouring columns (a column is 12(H+1)*8 bytes, not a multiple of
    // 128) are completed while they are still in L2.
    typedef double v2d __attribute__((ext_vector_type(2)));
    constexpr int kMaxPass = 4;  // rows per pass = 128; prefix_supported() keeps 12(H+1) <= 512
    const int lane = tid & 63, wave = tid >> 6, nwaves = bs >> 6;
    const int npass = (m + 127) / 128;
    for (int ls = wave; ls < ns; ls += nwaves) {
        const double *S = scen + ls * Y.total;
        double *out = A.jac + (size_t)(s0 + ls) * n * m;
        int kk[kMaxPass], gp[kMaxPass];
        bool is_pos[kMaxPass], live[kMaxPass];
        const double *src[kMaxPass];
#pragma unroll
        for (int ps = 0; ps < kMaxPass; ++ps) {
            const int r0 = 128 * ps + 2 * lane;
            live[ps] = ps < npass && r0 < m;
            kk[ps] = r0 / 12;
            const int g = (r0 - 12 * kk[ps]) >> 1;
            is_pos[ps] = g < 3;
            gp[ps] = is_pos[ps] ? g : g - 3;
            src[ps] = is_pos[ps] ? S + Y.jpos + 6 * kk[ps] + 2 * gp[ps] : S + Y.jvel + 2 * gp[ps];
        }
        for (int jj = 0; jj < H; ++jj) {
            const int tj = jj * H - (jj * (jj - 1)) / 2 - jj - 1;
            v2d val[kMaxPass];
#pragma unroll
            for (int ps = 0; ps < kMaxPass; ++ps) {
                val[ps] = v2d{0.0, 0.0};
                const bool hot = live[ps] && (is_pos[ps] ? (kk[ps] > jj) : (kk[ps] == jj + 1));
                if (hot) val[ps] = *reinterpret_cast<const v2d *>(src[ps] + 6 * (is_pos[ps] ? tj : jj));
            }
#pragma unroll
            for (int aa = 0; aa < 6; ++aa) {
                double *col = out + (size_t)(6 * jj + aa) * m + 2 * lane;
#pragma unroll
                for (int ps = 0; ps < kMaxPass; ++ps) {
                    if (ps < npass && live[ps]) {
                        const bool mine = gp[ps] == (aa >> 1);
                        const v2d v = (aa & 1) ? v2d{0.0, mine ? val[ps].y : 0.0} : v2d{mine ? val[ps].x : 0.0, 0.0};
                        *reinterpret_cast<v2d *>(col + 128 * ps) = v;  // plain store: nt measured 15 % slower (tools/store_bw.hip)
                    }
                }
            }
        }
    }
}

template <int MODE, bool SAME_DT, int MAXT, bool ROWS = false>
__global__ __launch_bounds__(MAXT) void prefix_kernel(const mpc_params P, const PrefixArgs A)
{
    extern __shared__ __align__(16) double lds[];
    prefix_body<MODE, SAME_DT, ROWS>(P, A, lds, (int)threadIdx.x, (int)blockDim.x, (int)blockIdx.x * A.S);
}

// kCompact's second half: one workgroup per (scenario, step jj) writes the six dense columns of that
// step (12(H+1) rows each; ONE contiguous run of 72(H+1) doubles) from the quotients kCompact left behind.
//
// A column is 6(H+1) 16-byte row pairs; thread t owns pairs t, t + 256, ... of EVERY column of the run.
// Pair (state kk, g) holds rows (2g, 2g+1) of that state: the position (g < 3) or velocity rows of joints
// (2gp, 2gp+1), non-zero only in columns aa = 2gp (first row) and 2gp+1 (second row), and only for kk > jj
// (position) or kk == jj+1 (velocity).  So a thread needs at most one 16-byte load per pair for all six columns,
// and all of them are issued and awaited BEFORE the first store: on gfx950 loads and stores share one counter, and
// a load whose value is awaited drains every store the wavefront has in flight (round 1's kernel loaded inside
// its store loop, with ~40 instructions of index arithmetic per store: 5.2 TB/s alone; this one 5.8-6.1).
// What was measured on the way (tools/h100_overlap.py): the rollouts overlap PERFECTLY with a plain
// fill of the same buffer (torch's: 9.5 ms || 14.2 ms = 15.1 ms), so the two kinds of work do not compete for
// anything fundamental; but beside THIS kernel at full occupancy they make no progress at all (the sum of the two
// times, whatever the stream priorities): every wave slot of the chip bursting 18 KB of stores makes the memory
// queues so deep that the fill itself slows down (19 ms) and each load and store of the rollouts waits behind them.
// Hence the dynamic-LDS request (MPC_EXPAND_LDS, unused by the code): it caps the expansion at three workgroups per
// CU, and at one beside two rollout workgroups.  A persistent grid (a few workgroups walking the runs) would be the
// direct way to cap it, and was tried in three forms: it never exceeded 3.7 TB/s (tools/fill_bw.hip mode 2 shows the
// same for a bare fill: multi-wave workgroups striding over private runs in lock step reach 4.0-4.7 TB/s).
//
// TMAJOR = false: `src` is the structured layout of the C-ABI ([6H velocity][3H(H+1) position quotients]);
// TMAJOR = true: the internal workspace layout [iteration t][phase-2 lane] written by prefix_kernel<kCompact>
// with A.tmajor: perturbation jj is the first (t from 0) or second (t from H - pi + 1) of the lane pair
// pi = min(jj, H-1-jj); iteration t0 + (kk-1-jj) holds the position quotient of state kk, t0 + (H-jj) the
// velocity quotient.
#ifndef MPC_PIPE_CHUNKS
#define MPC_PIPE_CHUNKS 16         // chunks a large long-horizon batch is pipelined in
#endif
constexpr int kExpandThreads = 256;
#ifndef MPC_EXPAND_LDS
#define MPC_EXPAND_LDS (48 * 1024) // dynamic LDS per expansion workgroup: unused by the kernel, it caps how many share a CU
#endif

// NP = passes per column = ceil(6(H+1) / 256), a template parameter so that the kernel holds NP loaded pairs and
// little else (24 VGPRs at NP = 3).
template <bool TMAJOR, int NP>
__global__ __launch_bounds__(kExpandThreads) void expand_kernel(int H, int count, const double *__restrict__ src,
                                                                int cstride, int items, double *__restrict__ jac)
{
    typedef double v2d __attribute__((ext_vector_type(2)));
    const int H1 = H + 1, m = 12 * H1;
    const unsigned bid = mpck::xcd_remap(blockIdx.x, gridDim.x);
    const int s = bid / H, jj = bid - s * H;
    if (s >= count) return;
    const double *cz = src + (size_t)s * cstride;
    const int tid = (int)threadIdx.x;
    const int pairs = 6 * H1;                         // 16-byte pairs per column
    // where state kk's position quotients (six joints, contiguous) and the velocity quotients of step jj start
    const int pi = (jj < H - 1 - jj) ? jj : H - 1 - jj;
    const int t0 = (jj == pi) ? 0 : H - pi + 1;
    const double *pos0 = TMAJOR ? cz + 6 * pi : cz + 6 * H + 6 * (jj * H - (jj * (jj - 1)) / 2 - jj - 1);
    const int pos_stride = TMAJOR ? items : 6;
    const int kadd = TMAJOR ? t0 - 1 - jj : 0;        // state kk's entry is pos0[(kk + kadd) * pos_stride + joint]
    const double *vel = TMAJOR ? cz + 6 * pi + (size_t)(t0 + H - jj) * items : cz + 6 * jj;

    v2d val[NP];
    int col0[NP];                                     // column of the pair's first row (the second row's is col0 + 1)
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int P = tid + kExpandThreads * i;
        const int kk = P / 6, g = P - 6 * kk;
        const bool is_pos = g < 3;
        const int gp = is_pos ? g : g - 3;
        const bool live = P < pairs;
        const bool hot = live && (is_pos ? (kk > jj) : (kk == jj + 1));
        col0[i] = live ? 2 * gp : -2;
        val[i] = v2d{0.0, 0.0};
        if (hot) val[i] = *reinterpret_cast<const v2d *>(is_pos ? pos0 + (size_t)(kk + kadd) * pos_stride + 2 * gp : vel + 2 * gp);
    }
    // every load has landed before the first store is issued: no wait for a load can then sit between two stores
#pragma unroll
    for (int i = 0; i < NP; ++i) asm volatile("" : "+v"(val[i].x), "+v"(val[i].y));
    double *out = jac + (size_t)bid * 6 * m + 2 * tid;   // run (s, jj) starts at (s n + 6 jj) m = bid * 6m
#pragma unroll
    for (int aa = 0; aa < 6; ++aa) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            if (col0[i] >= 0) {
                const bool mine = col0[i] == (aa & ~1);
                const v2d v = (aa & 1) ? v2d{0.0, mine ? val[i].y : 0.0} : v2d{mine ? val[i].x : 0.0, 0.0};
                *reinterpret_cast<v2d *>(out + aa * m + 2 * kExpandThreads * i) = v;
            }
        }
    }
}

template <bool TMAJOR, int NP>
hipError_t launch_expand_one(int H, int count, const double *src, int cstride, int items, double *jac, hipStream_t stream)
{
    if (MPC_EXPAND_LDS > 48 * 1024) {   // above the default dynamic-LDS limit
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(expand_kernel<TMAJOR, NP>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, MPC_EXPAND_LDS);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((expand_kernel<TMAJOR, NP>), dim3((unsigned)count * (unsigned)H), dim3(kExpandThreads), MPC_EXPAND_LDS,
                       stream, H, count, src, cstride, items, jac);
    return hipGetLastError();
}

template <bool TMAJOR>
hipError_t launch_expand_np(int H, int count, const double *src, int cstride, int items, double *jac, hipStream_t stream)
{
    switch ((6 * (H + 1) + kExpandThreads - 1) / kExpandThreads) {
    case 1: return launch_expand_one<TMAJOR, 1>(H, count, src, cstride, items, jac, stream);
    case 2: return launch_expand_one<TMAJOR, 2>(H, count, src, cstride, items, jac, stream);
    case 3: return launch_expand_one<TMAJOR, 3>(H, count, src, cstride, items, jac, stream);
    case 4: return launch_expand_one<TMAJOR, 4>(H, count, src, cstride, items, jac, stream);
    case 5: return launch_expand_one<TMAJOR, 5>(H, count, src, cstride, items, jac, stream);
    default: return hipErrorInvalidValue;
    }
}

bool ur_pattern(const mpc_params &P)
{
    const double ca[6] = {0, 1, 1, 0, 0, 1}, sa[6] = {1, 0, 0, 1, -1, 0};
    for (int i = 0; i < 6; ++i) {
        if (P.dh_ca[i] != ca[i] || P.dh_sa[i] != sa[i]) return false;
        const bool has_a = (i == 1 || i == 2), has_d = !has_a;
        if (!has_a && P.dh_a[i] != 0.0) return false;
        if (!has_d && P.dh_d[i] != 0.0) return false;
    }
    return true;
}

struct Geometry { int S, items, threads; size_t lds; };

Geometry geometry(const mpc_params &P, int mode)
{
    const int H = P.horizon, L = H + 1;
    const bool same_dt = P.dt_cost == P.dt_con;
    Geometry g;
    g.items = 6 * ((H + 1) / 2);
    const int lanes = (mode != kEval) ? (g.items > L ? g.items : L) : L;  // lanes one scenario needs
    g.S = lanes <= 64 ? 64 / lanes : 1;
    g.threads = ((g.S * lanes + 63) / 64) * 64;
    g.lds = sizeof(double) * ((size_t)((H + 1) & ~1) + (size_t)g.S * make_layout(H, mode, same_dt).total);
    return g;
}

bool fused_ok(const mpc_params &P)
{
    const Geometry g = geometry(P, kFused);
    return g.threads <= 128 && g.lds <= (size_t)kMaxLdsBytes && 12 * (P.horizon + 1) <= 512;
}

bool compact_ok(const mpc_params &P)
{
    const Geometry g = geometry(P, kCompact);
    return g.threads <= 512 && g.lds <= (size_t)kMaxLdsLarge;
}

template <int MODE, int MAXT, bool ROWS = false>
hipError_t launch_mode(const mpc_params &P, const PrefixArgs &A, const Geometry &g, unsigned grid, hipStream_t stream)
{
    const bool same_dt = P.dt_cost == P.dt_con;
    if (g.lds > 48 * 1024) {  // above the default dynamic-LDS limit
        hipError_t e = same_dt ? hipFuncSetAttribute(reinterpret_cast<const void *>(prefix_kernel<MODE, true, MAXT, ROWS>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds)
                               : hipFuncSetAttribute(reinterpret_cast<const void *>(prefix_kernel<MODE, false, MAXT, ROWS>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds);
        if (e != hipSuccess) return e;
    }
    if (same_dt) hipLaunchKernelGGL((prefix_kernel<MODE, true, MAXT, ROWS>), dim3(grid), dim3(g.threads), g.lds, stream, P, A);
    else hipLaunchKernelGGL((prefix_kernel<MODE, false, MAXT, ROWS>), dim3(grid), dim3(g.threads), g.lds, stream, P, A);
    return hipGetLastError();
}

}  // namespace

#ifndef MPC_TEAM_UNIT       // (kernels_team.hip includes this file for its device code only)
namespace mpck {

bool prefix_supported(const mpc_params &P) { return ur_pattern(P) && (fused_ok(P) || compact_ok(P)); }

size_t prefix_workspace_doubles_per_scenario(const mpc_params &P)
{
    if (!ur_pattern(P) || fused_ok(P) || !compact_ok(P)) return 0;
    // [H+3 iterations][6 ceil(H/2) phase-2 lanes]; equals the structured count 6H + 3H(H+1) for even H
    const size_t H = (size_t)P.horizon;
    return (H + 3) * 6 * ((H + 1) / 2);
}

size_t jacobian_nz_count(int horizon)
{
    const size_t H = (size_t)horizon;
    return 6 * H + 3 * H * (H + 1);
}

bool prefix_nz_supported(const mpc_params &P) { return ur_pattern(P) && compact_ok(P); }

hipError_t launch_prefix_nz(const mpc_params &P, const double *decay, const mpc_scenarios &sc, const double *x,
                            double *cost, double *grad, double *con, double *nz, hipStream_t stream)
{
    if (!prefix_nz_supported(P) || nz == nullptr) return hipErrorNotSupported;
    const Geometry g = geometry(P, kCompact);
    PrefixArgs A;
    A.decay = decay; A.sc = sc; A.x = x; A.cost = cost; A.grad = grad; A.con = con;
    A.jac = nz;      // kCompact only tests it against NULL ("the quotients are wanted")
    A.compact = nz; A.cstride = (int)jacobian_nz_count(P.horizon); A.tmajor = 0;
    A.S = g.S; A.items = g.items;
    A.list = nullptr; A.count_dev = nullptr;
    const unsigned grid = (unsigned)((sc.count + g.S - 1) / g.S);
    return g.threads <= 128 ? launch_mode<kCompact, 128>(P, A, g, grid, stream)
                            : launch_mode<kCompact, 512>(P, A, g, grid, stream);
}

hipError_t launch_prefix_rows(const mpc_params &P, const double *decay, const mpc_scenarios &sc, const double *x, double *cost,
                              double *grad, double *con, double *nz, const RowList &rows, hipStream_t stream)
{
    if (!prefix_nz_supported(P) || rows.list == nullptr || rows.count == nullptr) return hipErrorNotSupported;
    if (rows.bound <= 0) return hipSuccess;
    const bool fd = grad != nullptr || nz != nullptr;
    const int mode = fd ? kCompact : kEval;
    const Geometry g = geometry(P, mode);
    PrefixArgs A;
    A.decay = decay; A.sc = sc; A.x = x; A.cost = cost; A.grad = grad; A.con = con;
    A.jac = nz; A.compact = nz; A.cstride = (int)jacobian_nz_count(P.horizon); A.tmajor = 0;
    A.S = g.S; A.items = g.items;
    A.list = rows.list; A.count_dev = rows.count;
    const unsigned grid = (unsigned)((rows.bound + g.S - 1) / g.S);
    if (!fd)
        return g.threads <= 128 ? launch_mode<kEval, 128, true>(P, A, g, grid, stream)
                                : launch_mode<kEval, 512, true>(P, A, g, grid, stream);
    return g.threads <= 128 ? launch_mode<kCompact, 128, true>(P, A, g, grid, stream)
                            : launch_mode<kCompact, 512, true>(P, A, g, grid, stream);
}

// Test hook: the kernel's own sin/cos on an array, so its accuracy is checked directly and not only
// through the parity of whole evaluations (tests/test_gpu_parity.py::test_kernel_sincos_accuracy).
__global__ void sincos_probe_kernel(int n, const double *__restrict__ x, double *__restrict__ sn, double *__restrict__ cs)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) sincos_small(x[i], &sn[i], &cs[i]);
}

hipError_t launch_sincos_probe(int n, const double *x, double *sn, double *cs, hipStream_t stream)
{
    hipLaunchKernelGGL(sincos_probe_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, n, x, sn, cs);
    return hipGetLastError();
}

hipError_t launch_expand(const mpc_params &P, int count, const double *nz, double *jac, hipStream_t stream)
{
    return launch_expand_np<false>(P.horizon, count, nz, (int)jacobian_nz_count(P.horizon), 0, jac, stream);
}

hipError_t launch_prefix(const mpc_params &P, const double *decay, const mpc_scenarios &sc, const double *x,
                         double *cost, double *grad, double *con, double *jac, double *workspace,
                         size_t workspace_doubles, hipStream_t stream, const PrefixPipeline *pipe)
{
    if (!prefix_supported(P)) return hipErrorNotSupported;
    const bool fd = grad != nullptr || jac != nullptr;
    const bool fused = fused_ok(P);
    // the LDS image of kFused exists for the dense dc/dx; without it the leaner kCompact layout
    // (no staged quotients) lets more workgroups share a CU
    const int mode = !fd ? kEval : ((fused && jac != nullptr) ? kFused : kCompact);
    const Geometry g = geometry(P, mode);
    const int H = P.horizon, n = 6 * H, m = 12 * (H + 1);
    PrefixArgs A;
    A.decay = decay; A.sc = sc; A.x = x; A.cost = cost; A.grad = grad; A.con = con; A.jac = jac;
    A.compact = nullptr; A.cstride = 0; A.tmajor = 0;
    A.S = g.S; A.items = g.items;
    A.list = nullptr; A.count_dev = nullptr;
    if (mode != kCompact) {
        const unsigned grid = (unsigned)((sc.count + g.S - 1) / g.S);
        if (mode == kEval) return fused ? launch_mode<kEval, 128>(P, A, g, grid, stream) : launch_mode<kEval, 512>(P, A, g, grid, stream);
        return launch_mode<kFused, 128>(P, A, g, grid, stream);
    }
    if (jac == nullptr) {   // f, c, grad f only: one launch, nothing staged
        const unsigned grid = (unsigned)((sc.count + g.S - 1) / g.S);
        return g.threads <= 128 ? launch_mode<kCompact, 128>(P, A, g, grid, stream)
                                : launch_mode<kCompact, 512>(P, A, g, grid, stream);
    }
    // long horizon: scenarios go through the workspace in chunks (kernel A, then the expansion)
    const size_t per = prefix_workspace_doubles_per_scenario(P);
    int chunk = sc.count;
    bool piped = false;
    size_t half = 0;
    if (jac != nullptr) {
        if (workspace == nullptr || workspace_doubles < per) return hipErrorInvalidValue;
        // two halves of the workspace alternate, so that chunk i+1's rollouts (compute-bound, on `stream`)
        // overlap chunk i's expansion (store-bound, on pipe->aux)
        piped = pipe != nullptr && pipe->aux != nullptr && workspace_doubles / 2 >= per && sc.count >= 2048;
        half = piped ? workspace_doubles / 2 : workspace_doubles;
        const size_t fit = half / per;
        if (fit < (size_t)chunk) chunk = (int)fit;
        if (piped) {
            const int quarter = (sc.count + MPC_PIPE_CHUNKS - 1) / MPC_PIPE_CHUNKS;
            const int want = quarter > 1024 ? quarter : 1024;
            if (want < chunk) chunk = want;
        }
    }
    int used = 0;
    hipError_t err = hipSuccess;
    for (int first = 0, i = 0; first < sc.count && err == hipSuccess; first += chunk, ++i) {
        const int cnt = (sc.count - first < chunk) ? sc.count - first : chunk;
        const int h = piped ? (i & 1) : 0;
        PrefixArgs B = A;
        B.sc.count = cnt;
        B.sc.state0 += (size_t)first * 18; B.sc.target += (size_t)first * 6;
        B.sc.base_traj += (size_t)first * 12 * H; B.sc.Jc += (size_t)first * 36;
        if (B.sc.u_prev) B.sc.u_prev += (size_t)first * 6;
        if (B.sc.has_prev) B.sc.has_prev += first;
        B.x += (size_t)first * n; B.cost += first;
        if (B.grad) B.grad += (size_t)first * n;
        if (B.con) B.con += (size_t)first * m;
        if (B.jac) B.jac += (size_t)first * n * m;
        B.compact = workspace + (size_t)h * half; B.cstride = (int)per; B.tmajor = 1;
        if (piped && i >= 2) err = hipStreamWaitEvent(stream, pipe->drained[h], 0);   // half h is free again
        if (err == hipSuccess) err = launch_mode<kCompact, 512>(P, B, g, (unsigned)cnt, stream);
        if (err != hipSuccess || jac == nullptr) continue;
        hipStream_t xs = stream;
        if (piped) {
            err = hipEventRecord(pipe->filled[h], stream);
            if (err == hipSuccess) err = hipStreamWaitEvent(pipe->aux, pipe->filled[h], 0);
            if (err != hipSuccess) continue;
            xs = pipe->aux;
        }
        err = launch_expand_np<true>(H, cnt, B.compact, (int)per, g.items, B.jac, xs);
        if (err == hipSuccess && piped) err = hipEventRecord(pipe->drained[h], pipe->aux);
        if (err == hipSuccess && used < h + 1) used = h + 1;
    }
    // ordinary stream ordering for whatever the caller enqueues next; also on the error path, so that `stream` never
    // runs ahead of work already queued on the second stream
    for (int h = 0; piped && h < used; ++h) {
        const hipError_t e = hipStreamWaitEvent(stream, pipe->drained[h], 0);
        if (err == hipSuccess) err = e;
    }
    return err;
}

}  // namespace mpck
#endif  // MPC_TEAM_UNIT
