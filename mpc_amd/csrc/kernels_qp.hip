// SLSQP's QP subproblem on the device: one workgroup per problem runs the team-parallel LSQ of
// lsq_spmd.hpp (Kraft's LSQ -> LSI -> LDP -> NNLS).  Compiled with -ffp-contract=off: every entry is
// computed with the operations and in the order of the host core (csrc/slsqp_core.c), so the step and the
// multipliers come out bit-identical to the host's.  Workgroups are persistent over the batch (problem
// p = blockIdx.x, blockIdx.x + gridDim.x, ...), each with its own slice of the workspace.
#include "lsq_spmd.hpp"
#include "mpc_kernels.hpp"

namespace {

struct HipTeam {
    __device__ int tid() const { return (int)threadIdx.x; }
    __device__ int size() const { return (int)blockDim.x; }
    __device__ void sync() const { __syncthreads(); }
};

__global__ void __launch_bounds__(mpck::kQpThreads)
qp_kernel(int H, int count, const double *__restrict__ l, const double *__restrict__ g, const double *__restrict__ c,
          const double *__restrict__ nz, const double *__restrict__ lo, const double *__restrict__ hi, double *__restrict__ s,
          double *__restrict__ y, int32_t *__restrict__ mode, double *work, int *iwork, size_t wd, size_t wi)
{
    const int n = 6 * H, m = 12 * (H + 1), mg = m + 2 * n, nzc = 6 * H + 3 * H * (H + 1);
    const size_t ldl = (size_t)(n + 1) * n / 2 + 1;
    const lsqspmd::Work W = lsqspmd::carve(work + (size_t)blockIdx.x * wd, iwork + (size_t)blockIdx.x * wi, n, m, (int)blockDim.x);
    const HipTeam T;
    for (int p = (int)blockIdx.x; p < count; p += (int)gridDim.x) {
        const int md = lsqspmd::lsq(T, W, H, l + (size_t)p * ldl, g + (size_t)p * n, c + (size_t)p * m, nz + (size_t)p * nzc,
                                    lo + (size_t)p * n, hi + (size_t)p * n, s + (size_t)p * n, y + (size_t)p * mg);
        if (threadIdx.x == 0) mode[p] = md;
        __syncthreads();
    }
}

// One major-iteration step of an offloaded solve (slsqp_core.h: slsqp_state.offload) for the problems listed
// in `index`: reset or BFGS-update the problem's LDL' factor, solve the QP, keep the new multipliers and the
// Lagrangian gradient v = g - (dc/dx)^T r for the next update.  State rows are addressed by problem, the
// request's own arrays by position in the list.
__global__ void __launch_bounds__(mpck::kQpThreads)
qp_step_kernel(int H, int count, const int32_t *__restrict__ index, const int32_t *__restrict__ flags,
               const double *__restrict__ lo, const double *__restrict__ hi, const double *__restrict__ s_in, double *L,
               const double *__restrict__ G, const double *__restrict__ C, const double *__restrict__ NZ, double *R, double *V,
               double *__restrict__ s_out, double *__restrict__ r_out, int32_t *__restrict__ mode, double *work, int *iwork,
               size_t wd, size_t wi)
{
    const int n = 6 * H, m = 12 * (H + 1), mg = m + 2 * n, nzc = 6 * H + 3 * H * (H + 1);
    const size_t ldl = (size_t)(n + 1) * n / 2 + 1;
    const lsqspmd::Work W = lsqspmd::carve(work + (size_t)blockIdx.x * wd, iwork + (size_t)blockIdx.x * wi, n, m, (int)blockDim.x);
    const HipTeam T;
    for (int k = (int)blockIdx.x; k < count; k += (int)gridDim.x) {
        const size_t p = (size_t)index[k];
        double *l = L + p * ldl, *r = R + p * mg, *v = V + p * n;
        const double *g = G + p * n, *c = C + p * m, *nz = NZ + p * nzc;
        const int fl = flags[k];
        if (fl & 1) {                                            // the factor of the identity
            PAR_FOR(T, e, 0, (int)ldl - 1) l[e] = 0.0;
            T.sync();
            PAR_FOR(T, i, 0, n) l[i * n - (i * (i - 1)) / 2] = 1.0;
            T.sync();
        }
        if (fl & 2) lsqspmd::bfgs_update(T, W, H, l, g, nz, r, v, s_in + (size_t)k * n, W.xs);
        T.sync();
        const int md = lsqspmd::lsq(T, W, H, l, g, c, nz, lo + (size_t)k * n, hi + (size_t)k * n, s_out + (size_t)k * n, r);
        T.sync();
        if (md == 1) {
            PAR_FOR(T, i, 0, n) {
                const int j = i / 6, a = i % 6;
                const int tri = j * H - (j * (j - 1)) / 2 - j - 1;
                double t = 0.0;                                  // column i of dc/dx times r, rows ascending
                for (int kk = j + 1; kk <= H; ++kk) {
                    t = t + nz[6 * H + 6 * (tri + kk) + a] * r[12 * kk + a];
                    if (kk == j + 1) t = t + nz[i] * r[12 * (j + 1) + 6 + a];
                }
                v[i] = g[i] - t;
            }
            PAR_FOR(T, i, 0, m) r_out[(size_t)k * m + i] = r[i];
        }
        if (threadIdx.x == 0) mode[k] = md;
        __syncthreads();
    }
}

}  // namespace

namespace mpck {

hipError_t launch_qp_step(int H, int count, int teams, const int32_t *index, const int32_t *flags, const double *lo,
                          const double *hi, const double *s_in, double *L, const double *G, const double *C, const double *NZ,
                          double *R, double *V, double *s_out, double *r_out, int32_t *mode, double *work, int *iwork,
                          hipStream_t stream)
{
    hipLaunchKernelGGL(qp_step_kernel, dim3((unsigned)teams), dim3(kQpThreads), 0, stream, H, count, index, flags, lo, hi, s_in,
                       L, G, C, NZ, R, V, s_out, r_out, mode, work, iwork, qp_work_doubles(H), qp_work_ints(H));
    return hipGetLastError();
}


size_t qp_work_doubles(int H) { return lsqspmd::work_doubles(6 * H, 12 * (H + 1), kQpThreads); }
size_t qp_work_ints(int H) { return lsqspmd::work_ints(6 * H, 12 * (H + 1), kQpThreads); }

hipError_t launch_qp(int H, int count, int teams, const double *l, const double *g, const double *c, const double *nz,
                     const double *lo, const double *hi, double *s, double *y, int32_t *mode, double *work, int *iwork,
                     hipStream_t stream)
{
    hipLaunchKernelGGL(qp_kernel, dim3((unsigned)teams), dim3(kQpThreads), 0, stream, H, count, l, g, c, nz, lo, hi, s, y, mode,
                       work, iwork, qp_work_doubles(H), qp_work_ints(H));
    return hipGetLastError();
}

}  // namespace mpck
