// Per-scenario helper kernels either side of the hot path: Jacobian / forward kinematics of the
// arm (the MoveIt calls of the reference, DYN:82 and mpc_action_server.py:57-75) and the platform
// predictor (MPC:606-651).  One thread per scenario, rows contiguous per scenario.
// Compiled with -ffp-contract=off (same evaluation order as the CPU oracle).
#include "mpc_device.hpp"
#include "mpc_kernels.hpp"

namespace {

using namespace mpcdev;

__global__ void jacobian_kernel(const mpc_params P, int N, const double *__restrict__ q, double *__restrict__ Jout)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= N) return;
    double qq[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) qq[j] = q[(size_t)s * 6 + j];
    Chain C;
    ur_chain(P, qq, C);
    double J[6][6];
    chain_jacobian(C, J);
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = 0; c < 6; ++c) Jout[(size_t)s * 36 + 6 * r + c] = J[r][c];
}

__global__ void fk_kernel(const mpc_params P, int N, const double *__restrict__ q, double *__restrict__ pose)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= N) return;
    double qq[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) qq[j] = q[(size_t)s * 6 + j];
    Chain C;
    ur_chain(P, qq, C);
    // roll/pitch/yaw of R, static 'sxyz' axes (tf.transformations.euler_from_matrix)
    const double cy = sqrt(C.R[0][0] * C.R[0][0] + C.R[1][0] * C.R[1][0]);
    double roll, pitch, yaw;
    if (cy > 1e-12) {
        roll = atan2(C.R[2][1], C.R[2][2]);
        pitch = atan2(-C.R[2][0], cy);
        yaw = atan2(C.R[1][0], C.R[0][0]);
    } else {
        roll = atan2(-C.R[1][2], C.R[1][1]);
        pitch = atan2(-C.R[2][0], cy);
        yaw = 0.0;
    }
    double *out = pose + (size_t)s * 6;
    out[0] = C.pe[0]; out[1] = C.pe[1]; out[2] = C.pe[2];
    out[3] = roll; out[4] = pitch; out[5] = yaw;
}

struct BaseDyn { double mass, Iz, k_v, k_omega, tau_l, tau_a; };

// MPC:606-651; v_cmd / omega_cmd (cmd[0], cmd[1]) default to the current v / omega (MPC:624-625)
__device__ __forceinline__ void predict_base_dynamic_row(int H, double dt, const BaseDyn &D, const double *b, double *out,
                                                         const double *cmd = nullptr)
{
    double px = b[0], py = b[1], pz = b[2];
    double yaw = b[5], v = b[6], w = b[11];
    const double v_cmd = cmd ? cmd[0] : v, w_cmd = cmd ? cmd[1] : w;
    for (int k = 0; k < H; ++k) {
        const double force = (D.mass / D.tau_l) * (v_cmd - v);
        const double acc = (force - D.k_v * v) / D.mass;
        v = v + acc * dt;
        const double torque = (D.Iz / D.tau_a) * (w_cmd - w);
        const double alpha = (torque - D.k_omega * w) / D.Iz;
        w = w + alpha * dt;
        yaw = yaw + w * dt;
        double sn, cs;
        sincos(yaw, &sn, &cs);
        const double vx = v * cs, vy = v * sn;
        px = px + vx * dt; py = py + vy * dt; pz = pz + 0.0 * dt;
        double *r = out + (size_t)k * 12;
        r[0] = px; r[1] = py; r[2] = pz;
        r[3] = 0.0; r[4] = 0.0; r[5] = yaw;
        r[6] = vx; r[7] = vy; r[8] = 0.0;
        r[9] = 0.0; r[10] = 0.0; r[11] = w;
    }
}

__global__ void predict_base_dynamic_kernel(int N, int H, double dt, const BaseDyn D,
                                            const double *__restrict__ base, const double *__restrict__ cmd,
                                            double *__restrict__ out)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= N) return;
    predict_base_dynamic_row(H, dt, D, base + (size_t)s * 12, out + (size_t)s * H * 12, cmd ? cmd + (size_t)s * 2 : nullptr);
}

// R(psi) v of MPC:714-720
__device__ __forceinline__ void world_to_base(const double (&v)[3], double yaw, double (&out)[3])
{
    double sn, cs;
    sincos(yaw, &sn, &cs);
    out[0] = cs * v[0] + sn * v[1];
    out[1] = -sn * v[0] + cs * v[1];
    out[2] = v[2];
}

// Everything compute_control assembles before it calls minimize (MPC:104-185), one thread per controller.
__global__ void prepare_tick_kernel(const mpc_params P, int N, const BaseDyn D, const double *__restrict__ q,
                                    const double *__restrict__ dq, const double *__restrict__ ee_pose,
                                    const double *__restrict__ base, const double *__restrict__ prev_solution,
                                    const uint8_t *__restrict__ has_solution, double *__restrict__ state0,
                                    double *__restrict__ base_traj, double *__restrict__ Jc, double *__restrict__ u_init,
                                    double *__restrict__ x0, uint8_t *__restrict__ active)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= N) return;
    const int H = P.horizon, n = 6 * H;
    const double *b = base + (size_t)s * 12;
    // early exit of MPC:108: both platform velocities below 0.01
    const double nv = sqrt(b[6] * b[6] + b[7] * b[7] + b[8] * b[8]);
    const double nw = sqrt(b[9] * b[9] + b[10] * b[10] + b[11] * b[11]);
    active[s] = (nv < 0.01 && nw < 0.01) ? 0 : 1;
    double qq[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        qq[j] = q[(size_t)s * 6 + j];
        state0[(size_t)s * 18 + j] = qq[j];
        state0[(size_t)s * 18 + 6 + j] = dq != nullptr ? dq[(size_t)s * 6 + j] : 0.0;
        state0[(size_t)s * 18 + 12 + j] = ee_pose[(size_t)s * 6 + j];
    }
    predict_base_dynamic_row(H, P.dt_con, D, b, base_traj + (size_t)s * H * 12);
    Chain C;
    ur_chain(P, qq, C);
    double J[6][6];
    chain_jacobian(C, J);
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = 0; c < 6; ++c) Jc[(size_t)s * 36 + 6 * r + c] = J[r][c];
    // damped least-squares start (MPC:112-143,168-175): u = J^T (J J^T + 0.01 I)^-1 twist
    double r3[3], vrot[3], neg[3], twist[6], t3[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) r3[c] = ee_pose[(size_t)s * 6 + c] - b[c];
    vrot[0] = b[10] * r3[2] - b[11] * r3[1];
    vrot[1] = b[11] * r3[0] - b[9] * r3[2];
    vrot[2] = b[9] * r3[1] - b[10] * r3[0];
#pragma unroll
    for (int c = 0; c < 3; ++c) neg[c] = -(b[6 + c] + vrot[c]);
    world_to_base(neg, b[5], t3);
#pragma unroll
    for (int c = 0; c < 3; ++c) { twist[c] = t3[c]; neg[c] = -b[9 + c]; }
    world_to_base(neg, b[5], t3);
#pragma unroll
    for (int c = 0; c < 3; ++c) twist[3 + c] = t3[c];
    double M[6][7];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < 6; ++k) acc = acc + J[r][k] * J[c][k];
            M[r][c] = acc + (r == c ? 0.01 : 0.0);
        }
        M[r][6] = twist[r];
    }
    // Gaussian elimination with partial pivoting (what LAPACK's gesv does for numpy.linalg.solve)
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        int piv = k;
        double best = fabs(M[k][k]);
#pragma unroll
        for (int r = k + 1; r < 6; ++r)
            if (fabs(M[r][k]) > best) { best = fabs(M[r][k]); piv = r; }
#pragma unroll
        for (int c = 0; c < 7; ++c) {
            double top = M[k][c], other = top;
#pragma unroll
            for (int r = k + 1; r < 6; ++r) other = (piv == r) ? M[r][c] : other;
#pragma unroll
            for (int r = k + 1; r < 6; ++r) M[r][c] = (piv == r) ? top : M[r][c];
            M[k][c] = other;
        }
        const double inv = 1.0 / M[k][k];
#pragma unroll
        for (int r = k + 1; r < 6; ++r) {
            const double f = M[r][k] * inv;
#pragma unroll
            for (int c = k + 1; c < 7; ++c) M[r][c] = M[r][c] - f * M[k][c];
        }
    }
    double y[6];
#pragma unroll
    for (int r = 5; r >= 0; --r) {
        double acc = M[r][6];
#pragma unroll
        for (int c = r + 1; c < 6; ++c) acc = acc - M[r][c] * y[c];
        y[r] = acc / M[r][r];
    }
    double u[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        double acc = 0.0;
#pragma unroll
        for (int r = 0; r < 6; ++r) acc = acc + J[r][c] * y[r];
        u[c] = acc;
        u_init[(size_t)s * 6 + c] = acc;
    }
    // start point: shifted previous solution with the guess appended, else the guess repeated (MPC:177-185)
    if (x0 != nullptr) {
        const bool warm = prev_solution != nullptr && has_solution != nullptr && has_solution[s] != 0;
        for (int i = 0; i < n; ++i) {
            double v;
            if (warm && i < n - 6) v = prev_solution[(size_t)s * n + i + 6];
            else { const int c6 = i % 6; v = u[0]; for (int c = 1; c < 6; ++c) v = (c6 == c) ? u[c] : v; }
            x0[(size_t)s * n + i] = v;
        }
    }
}

}  // namespace

namespace mpck {

hipError_t launch_jacobian(const mpc_params &P, int N, const double *q, double *J, hipStream_t stream)
{
    hipLaunchKernelGGL(jacobian_kernel, dim3((N + 63) / 64), dim3(64), 0, stream, P, N, q, J);
    return hipGetLastError();
}

hipError_t launch_fk(const mpc_params &P, int N, const double *q, double *pose, hipStream_t stream)
{
    hipLaunchKernelGGL(fk_kernel, dim3((N + 63) / 64), dim3(64), 0, stream, P, N, q, pose);
    return hipGetLastError();
}

hipError_t launch_predict_base_dynamic(int N, int H, double dt, const double *base, const double *cmd, const double (&dyn)[6],
                                       double *out, hipStream_t stream)
{
    BaseDyn D{dyn[0], dyn[1], dyn[2], dyn[3], dyn[4], dyn[5]};
    hipLaunchKernelGGL(predict_base_dynamic_kernel, dim3((N + 63) / 64), dim3(64), 0, stream, N, H, dt, D, base, cmd,
                       out);
    return hipGetLastError();
}

// One workgroup per gathered row; the row's arrays are copied as one flat run of doubles.
__global__ void __launch_bounds__(128)
gather_scenarios_kernel(int H, mpc_scenarios src, const int32_t *__restrict__ index, mpc_scenarios dst)
{
    const int i = (int)blockIdx.x;
    const int r = index[i];
    const bool ok = r >= 0 && r < src.count;
    const double nan = __longlong_as_double(0x7ff8000000000000LL);
    const int nb = 12 * H;
    const bool prev = src.u_prev != nullptr && dst.u_prev != nullptr;
    const int total = 18 + 6 + 36 + nb + (prev ? 6 : 0);
    for (int f = (int)threadIdx.x; f < total; f += (int)blockDim.x) {
        if (f < 18)
            const_cast<double *>(dst.state0)[(size_t)i * 18 + f] = ok ? src.state0[(size_t)r * 18 + f] : nan;
        else if (f < 24)
            const_cast<double *>(dst.target)[(size_t)i * 6 + (f - 18)] = ok ? src.target[(size_t)r * 6 + (f - 18)] : nan;
        else if (f < 60)
            const_cast<double *>(dst.Jc)[(size_t)i * 36 + (f - 24)] = ok ? src.Jc[(size_t)r * 36 + (f - 24)] : nan;
        else if (f < 60 + nb)
            const_cast<double *>(dst.base_traj)[(size_t)i * nb + (f - 60)] = ok ? src.base_traj[(size_t)r * nb + (f - 60)] : nan;
        else
            const_cast<double *>(dst.u_prev)[(size_t)i * 6 + (f - 60 - nb)] = ok ? src.u_prev[(size_t)r * 6 + (f - 60 - nb)] : nan;
    }
    if (threadIdx.x == 0 && src.has_prev != nullptr && dst.has_prev != nullptr)
        const_cast<uint8_t *>(dst.has_prev)[i] = ok ? src.has_prev[r] : (uint8_t)0;
}

hipError_t launch_gather_scenarios(int H, const mpc_scenarios &src, const int32_t *index, int count,
                                   const mpc_scenarios &dst, hipStream_t stream)
{
    hipLaunchKernelGGL(gather_scenarios_kernel, dim3((unsigned)count), dim3(128), 0, stream, H, src, index, dst);
    return hipGetLastError();
}

hipError_t launch_prepare_tick(const mpc_params &P, int N, const double (&dyn)[6], const double *q, const double *dq,
                               const double *ee_pose, const double *base, const double *prev_solution,
                               const uint8_t *has_solution, double *state0, double *base_traj, double *Jc,
                               double *u_init, double *x0, uint8_t *active, hipStream_t stream)
{
    BaseDyn D{dyn[0], dyn[1], dyn[2], dyn[3], dyn[4], dyn[5]};
    hipLaunchKernelGGL(prepare_tick_kernel, dim3((N + 63) / 64), dim3(64), 0, stream, P, N, D, q, dq, ee_pose, base,
                       prev_solution, has_solution, state0, base_traj, Jc, u_init, x0, active);
    return hipGetLastError();
}

}  // namespace mpck
