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

}  // namespace

namespace mpck {

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
