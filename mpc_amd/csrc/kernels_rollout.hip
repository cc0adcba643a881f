// MPC_ALGO_ROLLOUT — the plain mapping: one workgroup per (scenario, perturbation group), one
// wavefront lane per horizon step (lane H takes the terminal cost), the scenario's control
// sequence, platform trajectory and tables staged in LDS, per-step terms combined across lanes
// with wavefront shuffles (a workgroup of several waves goes through LDS instead).
//
// Compiled with -ffp-contract=off: every product and sum rounds once, in the order of the numpy
// expressions of the reference (MPC:665-692 _cost_function, MPC:388-411 dynamics_constraint,
// DYN:101-128 predict_next_state), so this kernel differs from the CPU oracle only where the
// device's sin/cos differ from libm's in the last place.
//
// One rollout = H+1 lanes.  Lane k < H owns step k: q_k (sequential sum of u_j dt, j < k),
// J(q_k), the Euler increment J(q_k) u_k dt, and after the ordered prefix over the increments the
// stage cost L_k at (p_k, o_k).  Lane H owns q_H, J(q_H) and the terminal cost.  Rollout 0 of a
// scenario is the unperturbed one; rollout v+1 steps variable v by scipy's forward-difference
// step and also emits column v of dc/dx.
#include "mpc_device.hpp"
#include "mpc_kernels.hpp"

namespace {

using namespace mpcdev;

struct RolloutArgs {
    const double *decay;
    mpc_scenarios sc;
    const double *x;
    double *fvals;
    double *con;
    double *jac;
    int fstride;
    int n_pert;  // 0: unperturbed rollout only; n: plus one rollout per variable
    int G;       // workgroups per scenario
    int R;       // rollouts per workgroup
};

__device__ __forceinline__ double pick6(const double (&a)[6], int j)
{
    double v = a[0];
#pragma unroll
    for (int c = 1; c < 6; ++c) v = (j == c) ? a[c] : v;
    return v;
}

template <bool SINGLE_WAVE>
__global__ __launch_bounds__(SINGLE_WAVE ? 64 : 256) void rollout_kernel(const mpc_params P, const RolloutArgs A)
{
    extern __shared__ double lds[];
    const int H = P.horizon, n = 6 * H, m = 12 * (H + 1), L = H + 1;
    const int tid = threadIdx.x, bs = blockDim.x;
    const unsigned bid = mpck::xcd_remap(blockIdx.x, gridDim.x);
    const int s = bid / A.G, g = bid - s * A.G;

    double *xs = lds;          // [n]   control sequence of this scenario
    double *bt = xs + n;       // [12H] platform trajectory
    double *dec = bt + 12 * H; // [H]   0.95**k
    double *st0 = dec + H;     // [18]
    double *tgt = st0 + 18;    // [6]
    double *Jc = tgt + 6;      // [36]
    double *up = Jc + 36;      // [6]
    double *tb = up + 6;       // [bs*6] Euler increments   (multi-wave path)
    double *lb = tb + 6 * bs;  // [bs]   per-step costs      (multi-wave path)

    for (int i = tid; i < n; i += bs) xs[i] = A.x[(size_t)s * n + i];
    for (int i = tid; i < 12 * H; i += bs) bt[i] = A.sc.base_traj[(size_t)s * 12 * H + i];
    for (int i = tid; i < H; i += bs) dec[i] = A.decay[i];
    if (tid < 18) st0[tid] = A.sc.state0[(size_t)s * 18 + tid];
    if (tid < 6) tgt[tid] = A.sc.target[(size_t)s * 6 + tid];
    if (tid < 36) Jc[tid] = A.sc.Jc[(size_t)s * 36 + tid];
    const bool has_prev = A.sc.u_prev != nullptr && A.sc.has_prev != nullptr && A.sc.has_prev[s] != 0;
    if (tid < 6) up[tid] = has_prev ? A.sc.u_prev[(size_t)s * 6 + tid] : 0.0;
    __syncthreads();

    // ---- which rollout and which step this lane owns --------------------------------------
    int r, k;
    if (SINGLE_WAVE) { r = tid / L; k = tid - r * L; } else { r = 0; k = tid; }
    const int i = g * A.R + r;
    const bool valid = (r < A.R) && (i <= A.n_pert) && (k <= H);
    if (k > H) k = H;  // idle lanes compute a harmless copy; they never store
    const int var = (valid && i > 0) ? i - 1 : -1;
    const int vstep = (var >= 0) ? var / 6 : -1;
    const int vj = (var >= 0) ? var - 6 * vstep : -1;

    // scipy's step for this variable: x1[i] += h; dx = x1[i] - x0[i]  (NUMDIFF:594-596)
    double xv = 0.0, dx = 1.0;
    if (var >= 0) {
        const double x0v = xs[var];
        const double lim = pick6(P.v_lim, vj);
        xv = x0v + fd_step(x0v, -lim, lim, P.fd_eps);
        dx = xv - x0v;
    }

    // ---- this lane's control: u_k, or dq_H = u_{H-1} for the terminal lane (MPC:338) ------
    const int ku = (k < H) ? k : H - 1;
    double u[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) u[j] = (ku == vstep && j == vj) ? xv : xs[6 * ku + j];

    // ---- q_k by the reference's sequential updates q <- q + u*dt (DYN:105) ----------------
    double qc[6], qn[6];  // rollout of the cost (dt_cost) and of the constraint (dt_con)
#pragma unroll
    for (int j = 0; j < 6; ++j) { qc[j] = st0[j]; qn[j] = st0[j]; }
    double qb = (var >= 0) ? st0[vj] : 0.0;  // unperturbed constraint chain of the stepped joint
    for (int jj = 0; jj < H; ++jj) {
        if (jj < k) {
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const double uj = (jj == vstep && j == vj) ? xv : xs[6 * jj + j];
                qc[j] = qc[j] + uj * P.dt_cost;
                qn[j] = qn[j] + uj * P.dt_con;
            }
            if (var >= 0) qb = qb + xs[6 * jj + vj] * P.dt_con;
        }
    }

    // ---- constraint rows of state k (MPC:394-409) and column `var` of dc/dx ---------------
    if (valid && (A.con != nullptr || A.jac != nullptr)) {
        double dq[6];
#pragma unroll
        for (int j = 0; j < 6; ++j)
            dq[j] = (k == 0) ? st0[6 + j] : ((k - 1 == vstep && j == vj) ? xv : xs[6 * (k - 1) + j]);
        double c1[12];
#pragma unroll
        for (int j = 0; j < 6; ++j) { c1[j] = con_pos(P, j, qn[j]); c1[6 + j] = con_vel(P, j, dq[j]); }
        if (var < 0) {
            if (A.con != nullptr) {
                double2 *dst = reinterpret_cast<double2 *>(A.con + (size_t)s * m + 12 * k);
#pragma unroll
                for (int t = 0; t < 6; ++t) dst[t] = make_double2(c1[2 * t], c1[2 * t + 1]);
            }
        } else if (A.jac != nullptr) {
            // rows of the other joints are bit-identical in both rollouts: (c - c)/dx = +-0
            const double zero = copysign(0.0, dx);
            const double dqb = (k == 0) ? st0[6 + vj] : xs[6 * (k - 1) + vj];
            double col[12];
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                col[j] = (j == vj) ? (c1[j] - con_pos(P, j, qb)) / dx : zero;
                col[6 + j] = (j == vj) ? (c1[6 + j] - con_vel(P, j, dqb)) / dx : zero;
            }
            double2 *dst = reinterpret_cast<double2 *>(A.jac + ((size_t)s * n + var) * m + 12 * k);
#pragma unroll
            for (int t = 0; t < 6; ++t) dst[t] = make_double2(col[2 * t], col[2 * t + 1]);
        }
    }

    // ---- J(q_k) u and the Euler increment of the end-effector pose (DYN:109-120) ----------
    Chain C;
    ur_chain(P, qc, C);
    double w[6];
    chain_twist(C, u, w);
    double t[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) t[c] = w[c] * P.dt_cost;

    // ---- ordered prefix: p_k = p_0 + t_0 + ... + t_{k-1}, left to right -------------------
    double p[3] = {st0[12], st0[13], st0[14]};
    double o[3] = {st0[15], st0[16], st0[17]};
    const int rbase = r * L;
    if (SINGLE_WAVE) {
        for (int jj = 0; jj < H; ++jj) {
            double tj[6];
#pragma unroll
            for (int c = 0; c < 6; ++c) tj[c] = __shfl(t[c], rbase + jj);
            if (jj < k) {
#pragma unroll
                for (int c = 0; c < 3; ++c) { p[c] = p[c] + tj[c]; o[c] = o[c] + tj[3 + c]; }
            }
        }
    } else {
#pragma unroll
        for (int c = 0; c < 6; ++c) tb[6 * tid + c] = t[c];
        __syncthreads();
        for (int jj = 0; jj < k; ++jj) {
#pragma unroll
            for (int c = 0; c < 3; ++c) { p[c] = p[c] + tb[6 * jj + c]; o[c] = o[c] + tb[6 * jj + 3 + c]; }
        }
    }

    // ---- stage cost (lanes k < H) or terminal cost (lane H) -------------------------------
    double Lk;
    if (k < H) {
        double vl[3], va[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) { vl[c] = dot6(Jc + 6 * c, u); va[c] = dot6(Jc + 6 * (3 + c), u); }
        Lk = stage_cost(P, vl, va, u, p, o, bt + 12 * k, tgt, dec[k], k > 0 && has_prev, up, k > 0 && H > 1);
    } else {
        Lk = terminal_cost(P, w, p, o, bt + 12 * (H - 1), tgt);
    }

    // ---- total = ((L_0 + L_1) + ...) + L_T, the order of MPC:683,690 ----------------------
    double total = 0.0;
    if (SINGLE_WAVE) {
        for (int jj = 0; jj <= H; ++jj) total = total + __shfl(Lk, rbase + jj);
    } else {
        lb[tid] = Lk;
        __syncthreads();
        if (k == 0)
            for (int jj = 0; jj <= H; ++jj) total = total + lb[jj];
    }
    if (valid && k == 0) A.fvals[(size_t)s * A.fstride + i] = total;
}

__global__ void fd_finalize_kernel(const mpc_params P, int N, const double *__restrict__ x,
                                   const double *__restrict__ fvals, double *__restrict__ cost,
                                   double *__restrict__ grad)
{
    const int n = 6 * P.horizon;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)N * n) return;
    const int s = (int)(idx / n), v = (int)(idx - (size_t)s * n);
    const double f0 = fvals[(size_t)s * (n + 1)];
    if (v == 0) cost[s] = f0;
    if (grad != nullptr) {
        const double x0v = x[idx];
        const double lim = pick6(P.v_lim, v % 6);
        const double dx = (x0v + fd_step(x0v, -lim, lim, P.fd_eps)) - x0v;
        grad[idx] = (fvals[(size_t)s * (n + 1) + v + 1] - f0) / dx;
    }
}

}  // namespace

namespace mpck {

hipError_t launch_rollout(const mpc_params &P, const double *decay, const mpc_scenarios &sc, const double *x,
                          double *fvals, int fstride, double *con, double *jac, int n_pert,
                          hipStream_t stream)
{
    const int H = P.horizon, L = H + 1, n = 6 * H;
    RolloutArgs A;
    A.decay = decay; A.sc = sc; A.x = x; A.fvals = fvals; A.con = con; A.jac = jac;
    A.fstride = fstride; A.n_pert = n_pert;
    const bool single = L <= 64;
    A.R = single ? 64 / L : 1;
    A.G = (n_pert + 1 + A.R - 1) / A.R;
    const int bs = single ? 64 : ((L + 63) / 64) * 64;
    const size_t lds = sizeof(double) * (size_t)(n + 12 * H + H + 18 + 6 + 36 + 6 + (single ? 0 : 7 * bs));
    const unsigned grid = (unsigned)sc.count * (unsigned)A.G;
    if (single)
        hipLaunchKernelGGL(rollout_kernel<true>, dim3(grid), dim3(bs), lds, stream, P, A);
    else
        hipLaunchKernelGGL(rollout_kernel<false>, dim3(grid), dim3(bs), lds, stream, P, A);
    return hipGetLastError();
}

hipError_t launch_fd_finalize(const mpc_params &P, int N, const double *x, const double *fvals, double *cost,
                              double *grad, hipStream_t stream)
{
    const size_t total = (size_t)N * 6 * P.horizon;
    const unsigned grid = (unsigned)((total + 255) / 256);
    hipLaunchKernelGGL(fd_finalize_kernel, dim3(grid), dim3(256), 0, stream, P, N, x, fvals, cost, grad);
    return hipGetLastError();
}

}  // namespace mpck
