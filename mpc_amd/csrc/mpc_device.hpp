// Device-side building blocks shared by the gfx950 kernels of libmpc_amd.so.
//
// Every function states the reference lines it restates (MPC = ur_mpc.py, DYN = ur20_dynamics.py,
// NUMDIFF = scipy/optimize/_numdiff.py).  The arithmetic is written one operation per rounding in
// the order the reference's numpy expressions evaluate; a translation unit that wants that order
// kept compiles with -ffp-contract=off, one that does not lets hipcc fuse multiply-adds.
#pragma once

#include <hip/hip_runtime.h>

#include "mpc_amd.h"

namespace mpcdev {

// Joint axes z_i and origins o_i of the 6R chain in the Jacobian frame, plus the tip.
struct Chain {
    double z[6][3];
    double o[6][3];
    double R[3][3];
    double pe[3];
};

// One standard-DH link: Rz(theta) Tz(d) Tx(a) Rx(alpha) appended to (R, o).
__device__ __forceinline__ void dh_link(double (&R)[3][3], double (&o)[3], double c, double s, double a,
                                        double d, double ca, double sa)
{
    const double tx = a * c, ty = a * s, tz = d;
    const double m01 = -(s * ca), m02 = s * sa, m11 = c * ca, m12 = -(c * sa);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const double r0 = R[r][0], r1 = R[r][1], r2 = R[r][2];
        o[r] = o[r] + ((r0 * tx + r1 * ty) + r2 * tz);
        R[r][0] = r0 * c + r1 * s;
        R[r][1] = (r0 * m01 + r1 * m11) + r2 * sa;
        R[r][2] = (r0 * m02 + r1 * m12) + r2 * ca;
    }
}

// Walk the chain from precomputed sin/cos of the six joint angles.
__device__ __forceinline__ void ur_chain_sc(const mpc_params &P, const double (&sn)[6], const double (&cs)[6],
                                            Chain &C)
{
    double o[3] = {0.0, 0.0, 0.0};
    C.R[0][0] = P.base_c; C.R[0][1] = -P.base_s; C.R[0][2] = 0.0;
    C.R[1][0] = P.base_s; C.R[1][1] = P.base_c;  C.R[1][2] = 0.0;
    C.R[2][0] = 0.0;      C.R[2][1] = 0.0;       C.R[2][2] = 1.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            C.z[i][r] = C.R[r][2];
            C.o[i][r] = o[r];
        }
        dh_link(C.R, o, cs[i], sn[i], P.dh_a[i], P.dh_d[i], P.dh_ca[i], P.dh_sa[i]);
    }
#pragma unroll
    for (int r = 0; r < 3; ++r)
        C.pe[r] = o[r] + ((C.R[r][0] * P.tool[0] + C.R[r][1] * P.tool[1]) + C.R[r][2] * P.tool[2]);
}

__device__ __forceinline__ void ur_chain(const mpc_params &P, const double (&q)[6], Chain &C)
{
    double sn[6], cs[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) sincos(q[i], &sn[i], &cs[i]);
    ur_chain_sc(P, sn, cs, C);
}

// Geometric Jacobian (rows 0-2 linear, 3-5 angular; MPC:253): column i = [z_i x (pe - o_i); z_i].
// Stands in for MoveIt's getJacobian (DYN:82).
__device__ __forceinline__ void chain_jacobian(const Chain &C, double (&J)[6][6])
{
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const double z0 = C.z[i][0], z1 = C.z[i][1], z2 = C.z[i][2];
        const double d0 = C.pe[0] - C.o[i][0], d1 = C.pe[1] - C.o[i][1], d2 = C.pe[2] - C.o[i][2];
        J[0][i] = z1 * d2 - z2 * d1;
        J[1][i] = z2 * d0 - z0 * d2;
        J[2][i] = z0 * d1 - z1 * d0;
        J[3][i] = z0;
        J[4][i] = z1;
        J[5][i] = z2;
    }
}

// w = J(q) @ u without keeping J: each row's 6 products summed left to right (DYN:112).
__device__ __forceinline__ void chain_twist(const Chain &C, const double (&u)[6], double (&w)[6])
{
    double J[6][6];
    chain_jacobian(C, J);
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        double acc = J[r][0] * u[0];
#pragma unroll
        for (int c = 1; c < 6; ++c) acc = acc + J[r][c] * u[c];
        w[r] = acc;
    }
}

template <typename Ptr>
__device__ __forceinline__ double dot6(Ptr row, const double (&u)[6])
{
    double acc = row[0] * u[0];
#pragma unroll
    for (int c = 1; c < 6; ++c) acc = acc + row[c] * u[c];
    return acc;
}

__device__ __forceinline__ double sumsq3(double a, double b, double c) { return (a * a + b * b) + c * c; }

__device__ __forceinline__ double sumsq6(const double (&e)[6])
{
    double acc = e[0] * e[0];
#pragma unroll
    for (int c = 1; c < 6; ++c) acc = acc + e[c] * e[c];
    return acc;
}

// Compensation targets in the arm-base frame (stage MPC:263-283, terminal MPC:347-362).
// b = [P(3), Theta(3), V(3), Omega(3)]; R(psi) v with R = [[c,s,0],[-s,c,0],[0,0,1]] (MPC:714-720).
template <typename Ptr>
__device__ __forceinline__ void comp_targets(const double (&p)[3], Ptr b, double (&tl)[3], double (&ta)[3])
{
    const double r0 = p[0] - b[0], r1 = p[1] - b[1], r2 = p[2] - b[2];
    const double O0 = b[9], O1 = b[10], O2 = b[11];
    const double x0 = O1 * r2 - O2 * r1;
    const double x1 = O2 * r0 - O0 * r2;
    const double x2 = O0 * r1 - O1 * r0;
    const double w0 = -(b[6] + x0), w1 = -(b[7] + x1), w2 = -(b[8] + x2);
    const double a0 = -O0, a1 = -O1, a2 = -O2;
    double s, c;
    sincos(b[5], &s, &c);
    tl[0] = (c * w0 + s * w1) + 0.0 * w2;
    tl[1] = (-s * w0 + c * w1) + 0.0 * w2;
    tl[2] = (0.0 * w0 + 0.0 * w1) + 1.0 * w2;
    ta[0] = (c * a0 + s * a1) + 0.0 * a2;
    ta[1] = (-s * a0 + c * a1) + 0.0 * a2;
    ta[2] = (0.0 * a0 + 0.0 * a1) + 1.0 * a2;
}

// Stage cost L_k (MPC:249-328), on the state BEFORE u_k is applied (MPC:676-686).
// vl/va = Jc[:3]@u_k, Jc[3:]@u_k (MPC:260-261).
template <typename Ptr>
__device__ __forceinline__ double stage_cost(const mpc_params &P, const double (&vl)[3], const double (&va)[3],
                                             const double (&u)[6], const double (&p)[3], const double (&o)[3],
                                             Ptr b, Ptr tgt, double decay, bool damp, Ptr uprev, bool smooth)
{
    double tl[3], ta[3];
    comp_targets(p, b, tl, ta);
    const double c_lin = P.w_linear * decay * sumsq3(vl[0] - tl[0], vl[1] - tl[1], vl[2] - tl[2]);
    const double c_ang = P.w_angular * decay * sumsq3(va[0] - ta[0], va[1] - ta[1], va[2] - ta[2]);
    const double c_pos = P.w_position * decay * sumsq3(p[0] - tgt[0], p[1] - tgt[1], p[2] - tgt[2]);
    const double c_ori = P.w_orientation * decay * sumsq3(o[0] - tgt[3], o[1] - tgt[4], o[2] - tgt[5]);
    const double uu = sumsq6(u);
    const double c_ctl = P.w_control * uu;
    double c_damp = 0.0, c_smooth = 0.0;
    if (damp) {
        double du[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) du[j] = u[j] - uprev[j];
        c_damp = P.w_damping * sumsq6(du);
    }
    if (smooth) c_smooth = P.w_smooth * uu;
    return (((((c_lin + c_ang) + c_pos) + c_ori) + c_ctl) + c_damp) + c_smooth;
}

// Terminal cost L_T (MPC:330-381): w = J(q_H) @ dq_H, base = B_{H-1} (MPC:689).
template <typename Ptr>
__device__ __forceinline__ double terminal_cost(const mpc_params &P, const double (&w)[6], const double (&p)[3],
                                                const double (&o)[3], Ptr b, Ptr tgt)
{
    double tl[3], ta[3];
    comp_targets(p, b, tl, ta);
    const double c_lin = P.w_linear * P.terminal_factor * sumsq3(w[0] - tl[0], w[1] - tl[1], w[2] - tl[2]);
    const double c_ang = P.w_angular * P.terminal_factor * sumsq3(w[3] - ta[0], w[4] - ta[1], w[5] - ta[2]);
    const double c_pos = P.w_position * sumsq3(p[0] - tgt[0], p[1] - tgt[1], p[2] - tgt[2]);
    const double c_ori = P.w_orientation * sumsq3(o[0] - tgt[3], o[1] - tgt[4], o[2] - tgt[5]);
    return ((c_lin + c_ang) + c_pos) + c_ori;
}

// Forward-difference step of scipy's approx_derivative(method='2-point', abs_step=eps, bounds):
// NUMDIFF:498-524 then _adjust_scheme_to_bounds('1-sided') NUMDIFF:44-72.
__device__ __forceinline__ double fd_step(double x0, double lb, double ub, double eps)
{
    double h = eps;
    if ((x0 + h) - x0 == 0.0) {
        const double sgn = (x0 >= 0.0) ? 1.0 : -1.0;
        h = 1.4901161193847656e-08 * sgn * fmax(1.0, fabs(x0));  // sqrt(DBL_EPSILON)
    }
    const double lower = x0 - lb, upper = ub - x0;
    const double xs = x0 + h;
    const bool violated = (xs < lb) || (xs > ub);
    const bool fitting = fabs(h) <= fmax(lower, upper);
    if (violated && fitting) h = -h;
    if (!fitting) h = (upper >= lower) ? upper : -lower;
    return h;
}

// Two constraint rows of joint j in state k (MPC:398-406).
__device__ __forceinline__ double con_pos(const mpc_params &P, int j, double q)
{
    return fmin(q - P.q_lo[j], P.q_hi[j] - q);
}
__device__ __forceinline__ double con_vel(const mpc_params &P, int j, double dq) { return P.v_lim[j] - fabs(dq); }

}  // namespace mpcdev
