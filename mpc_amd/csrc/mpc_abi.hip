// C-ABI of libmpc_amd.so (declared in include/mpc_amd.h): context, argument checks, launches.
// No torch types, no exceptions across the boundary, nothing is computed on the host.
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>

#include <hip/hip_runtime.h>

#include "mpc_amd.h"
#include "mpc_kernels.hpp"

#ifndef MPC_SOLVE_OVERLAP
#define MPC_SOLVE_OVERLAP 1      // 0: a round's evaluations and QPs on one stream, one after the other (development)
#endif
constexpr int kSolveOverlapFrom = 128;   // mpc_solve_rounds: from this many unfinished problems on, evaluations run beside QPs

struct mpc_ctx {
    int device = 0;
    int horizon = 0;
    int max_scenarios = 0;
    bool have_params = false;
    mpc_params params{};
    double *d_decay = nullptr;  // [H]
    double *d_fvals = nullptr;  // [max_scenarios][n+1], MPC_ALGO_ROLLOUT only
    double *d_work = nullptr;   // long-horizon MPC_ALGO_PREFIX: FD quotients between its two kernels
    size_t work_doubles = 0;
    mpck::PrefixPipeline pipe;  // second stream + events: overlaps the two long-horizon kernels
    double *d_qp_work = nullptr;  // mpc_qp_solve: workspace of the resident workgroups (allocated on first use)
    int *d_qp_iwork = nullptr;
    int qp_teams = 0;
    int32_t *aug_seen = nullptr;      // pinned host memory the device writes: a resident solve met an augmented QP request
    const void *solve_ws = nullptr;   // mpc_solve_*: the workspace mpc_solve_begin set up, and how many rounds have run on
    int solve_round = 0;              // it: round r's requests are in list buffer (r + 1) % 2, counter slot (r + 1) % 3
    hipStream_t solve_side = nullptr; // mpc_solve_rounds: a round's evaluations run here, beside its QPs (made by mpc_solve_begin)
    hipEvent_t solve_forked[2] = {nullptr, nullptr}, solve_joined[2] = {nullptr, nullptr};
    double *d_static = nullptr;   // this device's copy of the file-scope `scratch` array (mpc_eval_fd_static)
    char err[512] = {0};
};

// mpc_eval_fd_static: the per-rollout costs of MPC_ALGO_ROLLOUT (what mpc_ctx_create otherwise allocates as
// d_fvals) live in this file-scope device array instead, so that entry point allocates nothing.  The name is
// a host-side stand-in: its device address is taken once per device (static_scratch_for) and only that
// pointer is ever handed to a kernel.  Each GPU the library runs on holds its own 16 MiB copy.
static __device__ __align__(256) double scratch[MPC_STATIC_SCRATCH_DOUBLES];

namespace {

thread_local char g_err[512] = {0};

constexpr int kMaxDevices = 64;
std::mutex g_scratch_mutex;
double *g_scratch_ptr[kMaxDevices] = {nullptr};

// Device address of `scratch` on `device`; resolved once per device, with that device current.
hipError_t static_scratch_for(int device, double **out)
{
    *out = nullptr;
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        hipError_t e = hipSetDevice(device);
        if (e != hipSuccess) return e;
        double *ptr = nullptr;
        e = hipGetSymbolAddress(reinterpret_cast<void **>(&ptr), HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        if (!ptr) return hipErrorInvalidSymbol;
        g_scratch_ptr[device] = ptr;
    }
    *out = g_scratch_ptr[device];
    return hipSuccess;
}

constexpr size_t kWorkCapBytes = size_t(2) << 30;

int fail(mpc_ctx *ctx, int code, const char *fmt, ...)
{
    char *dst = ctx ? ctx->err : g_err;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(dst, 512, fmt, ap);
    va_end(ap);
    return code;
}

int hip_fail(mpc_ctx *ctx, hipError_t e, const char *what)
{
    // the error is reported through the return code and mpc_last_error; the runtime's own "last error" is cleared,
    // or the caller's next launch check (torch's, say) would trip over an error that was already handled here
    (void)hipGetLastError();
    return fail(ctx, MPC_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
}

#define MPC_HIP(ctx, call)                                   \
    do {                                                     \
        hipError_t e_ = (call);                              \
        if (e_ != hipSuccess) return hip_fail(ctx, e_, #call); \
    } while (0)

int check_ready(mpc_ctx *ctx)
{
    if (!ctx) return fail(nullptr, MPC_ERR_INVALID_ARG, "ctx is NULL");
    if (!ctx->have_params) return fail(ctx, MPC_ERR_NO_PARAMS, "mpc_set_params has not been called");
    return MPC_OK;
}

int check_scenarios(mpc_ctx *ctx, const mpc_scenarios *sc, const double *x)
{
    if (!sc) return fail(ctx, MPC_ERR_INVALID_ARG, "scenarios is NULL");
    if (sc->count < 0 || sc->count > ctx->max_scenarios)
        return fail(ctx, MPC_ERR_INVALID_ARG, "scenario count %d outside [0, %d]", sc->count, ctx->max_scenarios);
    if (sc->count == 0) return MPC_OK;
    if (!sc->state0 || !sc->target || !sc->base_traj || !sc->Jc || !x)
        return fail(ctx, MPC_ERR_INVALID_ARG, "state0/target/base_traj/Jc/x must be non-NULL");
    if (sc->has_prev && !sc->u_prev) return fail(ctx, MPC_ERR_INVALID_ARG, "has_prev given without u_prev");
    return MPC_OK;
}

int resolve_algo(const mpc_ctx *ctx)
{
    int a = ctx->params.algo;
    if (a == MPC_ALGO_AUTO) a = mpck::prefix_supported(ctx->params) ? MPC_ALGO_PREFIX : MPC_ALGO_ROLLOUT;
    return a;
}

}  // namespace

extern "C" {

int mpc_abi_version(void) { return MPC_AMD_ABI_VERSION; }

const char *mpc_last_error(const mpc_ctx *ctx) { return ctx ? ctx->err : g_err; }

int mpc_ctx_create(int device, int horizon, int max_scenarios, mpc_ctx **out)
{
    if (!out) return fail(nullptr, MPC_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    if (horizon <= 0) return fail(nullptr, MPC_ERR_INVALID_ARG, "Horizon must be positive");  // MPC:33-34
    if (horizon > MPC_MAX_HORIZON)
        return fail(nullptr, MPC_ERR_UNSUPPORTED, "horizon %d > MPC_MAX_HORIZON %d", horizon, MPC_MAX_HORIZON);
    if (max_scenarios <= 0) return fail(nullptr, MPC_ERR_INVALID_ARG, "max_scenarios must be positive");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(nullptr, MPC_ERR_NO_DEVICE, "no HIP device visible: this library has no CPU path");
    if (device < 0 || device >= ndev)
        return fail(nullptr, MPC_ERR_INVALID_ARG, "device %d outside [0, %d)", device, ndev);
    mpc_ctx *ctx = new (std::nothrow) mpc_ctx();
    if (!ctx) return fail(nullptr, MPC_ERR_HIP, "out of host memory");
    ctx->device = device;
    ctx->horizon = horizon;
    ctx->max_scenarios = max_scenarios;
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = static_scratch_for(device, &ctx->d_static);   // set-up: once per device
    if (e == hipSuccess) e = hipMalloc(&ctx->d_decay, sizeof(double) * (size_t)horizon);
    // (the per-rollout costs of MPC_ALGO_ROLLOUT are allocated by mpc_set_params, and only if that kernel will run)
    if (e != hipSuccess) {
        int rc = hip_fail(nullptr, e, "mpc_ctx_create");
        mpc_ctx_destroy(ctx);
        return rc;
    }
    *out = ctx;
    return MPC_OK;
}

void mpc_ctx_destroy(mpc_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->d_decay) (void)hipFree(ctx->d_decay);
    if (ctx->d_fvals) (void)hipFree(ctx->d_fvals);
    if (ctx->d_work) (void)hipFree(ctx->d_work);
    if (ctx->aug_seen) (void)hipHostFree(ctx->aug_seen);
    for (int h = 0; h < 2; ++h) {
        if (ctx->solve_forked[h]) (void)hipEventDestroy(ctx->solve_forked[h]);
        if (ctx->solve_joined[h]) (void)hipEventDestroy(ctx->solve_joined[h]);
    }
    if (ctx->solve_side) (void)hipStreamDestroy(ctx->solve_side);
    if (ctx->d_qp_work) (void)hipFree(ctx->d_qp_work);
    if (ctx->d_qp_iwork) (void)hipFree(ctx->d_qp_iwork);
    for (int h = 0; h < 2; ++h) {
        if (ctx->pipe.filled[h]) (void)hipEventDestroy(ctx->pipe.filled[h]);
        if (ctx->pipe.drained[h]) (void)hipEventDestroy(ctx->pipe.drained[h]);
    }
    if (ctx->pipe.aux) (void)hipStreamDestroy(ctx->pipe.aux);
    delete ctx;
}

int mpc_set_params(mpc_ctx *ctx, const mpc_params *p, const double *decay_host, int n_decay)
{
    if (!ctx) return fail(nullptr, MPC_ERR_INVALID_ARG, "ctx is NULL");
    if (!p || !decay_host) return fail(ctx, MPC_ERR_INVALID_ARG, "params/decay is NULL");
    if (p->horizon != ctx->horizon)
        return fail(ctx, MPC_ERR_INVALID_ARG, "params.horizon %d != context horizon %d", p->horizon, ctx->horizon);
    if (n_decay != ctx->horizon) return fail(ctx, MPC_ERR_INVALID_ARG, "decay table must have H entries");
    if (!(p->dt_cost > 0.0) || !(p->dt_con > 0.0))
        return fail(ctx, MPC_ERR_INVALID_ARG, "Time step must be positive");  // MPC:35-36
    if (!(p->fd_eps > 0.0)) return fail(ctx, MPC_ERR_INVALID_ARG, "fd_eps must be positive");
    if (p->algo < MPC_ALGO_AUTO || p->algo > MPC_ALGO_PREFIX)
        return fail(ctx, MPC_ERR_INVALID_ARG, "unknown algo %d", p->algo);
    for (int j = 0; j < 6; ++j) {
        if (!(p->q_lo[j] <= p->q_hi[j]) || !(p->v_lim[j] > 0.0))
            return fail(ctx, MPC_ERR_INVALID_ARG, "joint %d: limits are not ordered / positive", j);
    }
    const double *all = reinterpret_cast<const double *>(&p->dt_cost);
    const size_t ndbl = (sizeof(mpc_params) - offsetof(mpc_params, dt_cost)) / sizeof(double);
    for (size_t i = 0; i < ndbl; ++i)
        if (!std::isfinite(all[i])) return fail(ctx, MPC_ERR_INVALID_ARG, "params contain a non-finite value");
    if (p->algo == MPC_ALGO_PREFIX && !mpck::prefix_supported(*p))
        return fail(ctx, MPC_ERR_UNSUPPORTED, "MPC_ALGO_PREFIX does not support these parameters");
    MPC_HIP(ctx, hipSetDevice(ctx->device));
    MPC_HIP(ctx, hipMemcpy(ctx->d_decay, decay_host, sizeof(double) * (size_t)n_decay, hipMemcpyHostToDevice));
    // long horizons: workspace for the FD quotients, at most kWorkCapBytes (batches go through it in chunks)
    const size_t per = (p->algo != MPC_ALGO_ROLLOUT) ? mpck::prefix_workspace_doubles_per_scenario(*p) : 0;
    size_t want = per * (size_t)ctx->max_scenarios;
    const size_t cap = kWorkCapBytes / sizeof(double);
    if (want > cap) want = (cap / per) * per;
    if (want < per) want = per;
    if (per == 0) want = 0;
    if (want != ctx->work_doubles) {
        if (ctx->d_work) (void)hipFree(ctx->d_work);
        ctx->d_work = nullptr;
        ctx->work_doubles = 0;
        if (want) MPC_HIP(ctx, hipMalloc(&ctx->d_work, want * sizeof(double)));
        ctx->work_doubles = want;
    }
    if (want && !ctx->pipe.aux) {
        // the expansion (a store-bound fill) runs here beside the next chunk's rollouts (the priority of either stream
        // made no measurable difference: what balances the two kernels is the expansion's occupancy cap, kernels_prefix.hip)
        int prio_low = 0, prio_high = 0;
        MPC_HIP(ctx, hipDeviceGetStreamPriorityRange(&prio_low, &prio_high));
        MPC_HIP(ctx, hipStreamCreateWithPriority(&ctx->pipe.aux, hipStreamNonBlocking, prio_high));
        for (int h = 0; h < 2; ++h) {
            MPC_HIP(ctx, hipEventCreateWithFlags(&ctx->pipe.filled[h], hipEventDisableTiming));
            MPC_HIP(ctx, hipEventCreateWithFlags(&ctx->pipe.drained[h], hipEventDisableTiming));
        }
    }
    // per-rollout costs between the two kernels of MPC_ALGO_ROLLOUT: only that algorithm reads them
    const bool rollout = p->algo == MPC_ALGO_ROLLOUT || (p->algo == MPC_ALGO_AUTO && !mpck::prefix_supported(*p));
    if (rollout && !ctx->d_fvals) {
        MPC_HIP(ctx, hipMalloc(&ctx->d_fvals, sizeof(double) * (size_t)ctx->max_scenarios * (size_t)(6 * ctx->horizon + 1)));
    } else if (!rollout && ctx->d_fvals) {
        (void)hipFree(ctx->d_fvals);
        ctx->d_fvals = nullptr;
    }
    ctx->params = *p;
    ctx->have_params = true;
    return MPC_OK;
}

int mpc_eval(mpc_ctx *ctx, const mpc_scenarios *sc, const double *x, double *cost, double *con, void *stream)
{
    int rc = check_ready(ctx);
    if (rc) return rc;
    rc = check_scenarios(ctx, sc, x);
    if (rc) return rc;
    if (sc->count == 0) return MPC_OK;
    if (!cost) return fail(ctx, MPC_ERR_INVALID_ARG, "cost is NULL");
    MPC_HIP(ctx, hipSetDevice(ctx->device));
    if (resolve_algo(ctx) == MPC_ALGO_PREFIX) {  // phase 1 of the prefix kernel: bit-identical f to mpc_eval_fd's
        MPC_HIP(ctx, mpck::launch_prefix(ctx->params, ctx->d_decay, *sc, x, cost, nullptr, con, nullptr, nullptr, 0,
                                         static_cast<hipStream_t>(stream)));
        return MPC_OK;
    }
    MPC_HIP(ctx, mpck::launch_rollout(ctx->params, ctx->d_decay, *sc, x, cost, 1, con, nullptr, 0,
                                      static_cast<hipStream_t>(stream)));
    return MPC_OK;
}

int mpc_eval_fd(mpc_ctx *ctx, const mpc_scenarios *sc, const double *x, double *cost, double *grad, double *con,
                double *con_jac, void *stream)
{
    int rc = check_ready(ctx);
    if (rc) return rc;
    rc = check_scenarios(ctx, sc, x);
    if (rc) return rc;
    if (sc->count == 0) return MPC_OK;
    if (!cost) return fail(ctx, MPC_ERR_INVALID_ARG, "cost is NULL");
    hipStream_t st = static_cast<hipStream_t>(stream);
    MPC_HIP(ctx, hipSetDevice(ctx->device));
    if (resolve_algo(ctx) == MPC_ALGO_PREFIX) {
        MPC_HIP(ctx, mpck::launch_prefix(ctx->params, ctx->d_decay, *sc, x, cost, grad, con, con_jac, ctx->d_work,
                                         ctx->work_doubles, st, ctx->pipe.aux ? &ctx->pipe : nullptr));
        return MPC_OK;
    }
    const int n = 6 * ctx->horizon;
    MPC_HIP(ctx, mpck::launch_rollout(ctx->params, ctx->d_decay, *sc, x, ctx->d_fvals, n + 1, con, con_jac, n, st));
    MPC_HIP(ctx, mpck::launch_fd_finalize(ctx->params, sc->count, x, ctx->d_fvals, cost, grad, st));
    return MPC_OK;
}

int mpc_eval_fd_static(mpc_ctx *ctx, const mpc_scenarios *sc, const double *x, double *cost, double *grad, double *con,
                       double *con_jac, void *stream)
{
    int rc = check_ready(ctx);
    if (rc) return rc;
    rc = check_scenarios(ctx, sc, x);
    if (rc) return rc;
    if (sc->count == 0) return MPC_OK;
    if (!cost) return fail(ctx, MPC_ERR_INVALID_ARG, "cost is NULL");
    const int n = 6 * ctx->horizon;
    if ((size_t)sc->count * (size_t)(n + 1) > (size_t)MPC_STATIC_SCRATCH_DOUBLES)
        return fail(ctx, MPC_ERR_INVALID_ARG,
                    "mpc_eval_fd_static: %d scenarios x (6H+1 = %d) rollout costs exceed the static scratch of %d doubles "
                    "(at most %d scenarios at this horizon)", sc->count, n + 1, (int)MPC_STATIC_SCRATCH_DOUBLES,
                    (int)(MPC_STATIC_SCRATCH_DOUBLES / (n + 1)));
    double *fvals = ctx->d_static;   // the device address, never the array's name
    if (!fvals) return fail(ctx, MPC_ERR_HIP, "static scratch was not resolved on device %d", ctx->device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    MPC_HIP(ctx, hipSetDevice(ctx->device));
    // every fvals[s][0..n] that fd_finalize reads is written by this launch: nothing has to start at zero
    MPC_HIP(ctx, mpck::launch_rollout(ctx->params, ctx->d_decay, *sc, x, fvals, n + 1, con, con_jac, n, st));
    MPC_HIP(ctx, mpck::launch_fd_finalize(ctx->params, sc->count, x, fvals, cost, grad, st));
    return MPC_OK;
}

int mpc_jacobian_nz_count(int horizon)
{
    return (horizon > 0 && horizon <= MPC_MAX_HORIZON) ? (int)mpck::jacobian_nz_count(horizon) : 0;
}

int mpc_eval_fd_nz(mpc_ctx *ctx, const mpc_scenarios *sc, const double *x, double *cost, double *grad, double *con,
                   double *con_jac_nz, void *stream)
{
    int rc = check_ready(ctx);
    if (rc) return rc;
    if (resolve_algo(ctx) != MPC_ALGO_PREFIX || !mpck::prefix_nz_supported(ctx->params))  // also for an empty batch: a probe
        return fail(ctx, MPC_ERR_UNSUPPORTED, "the structured dc/dx needs MPC_ALGO_PREFIX (UR twist pattern, H <= 170)");
    rc = check_scenarios(ctx, sc, x);
    if (rc) return rc;
    if (sc->count == 0) return MPC_OK;
    if (!cost || !con_jac_nz) return fail(ctx, MPC_ERR_INVALID_ARG, "cost / con_jac_nz is NULL");
    MPC_HIP(ctx, hipSetDevice(ctx->device));
    MPC_HIP(ctx, mpck::launch_prefix_nz(ctx->params, ctx->d_decay, *sc, x, cost, grad, con, con_jac_nz,
                                        static_cast<hipStream_t>(stream)));
    return MPC_OK;
}

int mpc_expand_jacobian(mpc_ctx *ctx, int count, const double *con_jac_nz, double *con_jac, void *stream)
{
    int rc = check_ready(ctx);
    if (rc) return rc;
    if (count < 0 || count > ctx->max_scenarios)
        return fail(ctx, MPC_ERR_INVALID_ARG, "count %d outside [0, %d]", count, ctx->max_scenarios);
    if (count == 0) return MPC_OK;
    if (!con_jac_nz || !con_jac) return fail(ctx, MPC_ERR_INVALID_ARG, "con_jac_nz / con_jac is NULL");
    MPC_HIP(ctx, hipSetDevice(ctx->device));
    MPC_HIP(ctx, mpck::launch_expand(ctx->params, count, con_jac_nz, con_jac, static_cast<hipStream_t>(stream)));
    return MPC_OK;
}

int mpc_jacobian(mpc_ctx *ctx, int count, const double *q, double *J, void *stream)
{
    int rc = check_ready(ctx);
    if (rc) return rc;
    if (count < 0 || (count > 0 && (!q || !J))) return fail(ctx, MPC_ERR_INVALID_ARG, "bad count / NULL q or J");
    if (count == 0) return MPC_OK;
    MPC_HIP(ctx, hipSetDevice(ctx->device));
    MPC_HIP(ctx, mpck::launch_jacobian(ctx->params, count, q, J, static_cast<hipStream_t>(stream)));
    return MPC_OK;
}

int mpc_fk(mpc_ctx *ctx, int count, const double *q, double *pose, void *stream)
{
    int rc = check_ready(ctx);
    if (rc) return rc;
    if (count < 0 || (count > 0 && (!q || !pose))) return fail(ctx, MPC_ERR_INVALID_ARG, "bad count / NULL q or pose");
    if (count == 0) return MPC_OK;
    MPC_HIP(ctx, hipSetDevice(ctx->device));
    MPC_HIP(ctx, mpck::launch_fk(ctx->params, count, q, pose, static_cast<hipStream_t>(stream)));
    return MPC_OK;
}

int mpc_sincos_probe(mpc_ctx *ctx, int count, const double *x, double *sn, double *cs, void *stream)
{
    if (!ctx) return MPC_ERR_INVALID_ARG;
    if (count < 0 || (count > 0 && (!x || !sn || !cs))) return fail(ctx, MPC_ERR_INVALID_ARG, "bad count / NULL x, sn or cs");
    if (count == 0) return MPC_OK;
    MPC_HIP(ctx, hipSetDevice(ctx->device));
    MPC_HIP(ctx, mpck::launch_sincos_probe(count, x, sn, cs, static_cast<hipStream_t>(stream)));
    return MPC_OK;
}

int mpc_ordered_sum_probe(mpc_ctx *ctx, int count, const double *terms, const double *t0, const int32_t *first, double *out,
                          void *stream)
{
    if (!ctx) return MPC_ERR_INVALID_ARG;
    if (count < 0 || (count > 0 && (!terms || !t0 || !first || !out)))
        return fail(ctx, MPC_ERR_INVALID_ARG, "bad count / NULL terms, t0, first or out");
    if (count == 0) return MPC_OK;
    MPC_HIP(ctx, hipSetDevice(ctx->device));
    MPC_HIP(ctx, mpck::launch_ordered_sum_probe(count, terms, t0, first, out, static_cast<hipStream_t>(stream)));
    return MPC_OK;
}

int mpc_predict_base_dynamic(mpc_ctx *ctx, int count, const double *base, const double *dyn_host,
                             double *base_traj, void *stream)
{
    return mpc_predict_base_dynamic_cmd(ctx, count, base, nullptr, dyn_host, base_traj, stream);
}

int mpc_predict_base_dynamic_cmd(mpc_ctx *ctx, int count, const double *base, const double *cmd, const double *dyn_host,
                                 double *base_traj, void *stream)
{
    int rc = check_ready(ctx);
    if (rc) return rc;
    if (count < 0 || !dyn_host || (count > 0 && (!base || !base_traj)))
        return fail(ctx, MPC_ERR_INVALID_ARG, "bad count / NULL base, dyn or base_traj");
    for (int i = 0; i < 6; ++i)
        if (!std::isfinite(dyn_host[i]) || (i != 2 && i != 3 && !(dyn_host[i] > 0.0)))
            return fail(ctx, MPC_ERR_INVALID_ARG, "base dynamics: mass, I_z and the lags must be positive");
    if (count == 0) return MPC_OK;
    double dyn[6];
    std::memcpy(dyn, dyn_host, sizeof(dyn));
    MPC_HIP(ctx, hipSetDevice(ctx->device));
    MPC_HIP(ctx, mpck::launch_predict_base_dynamic(count, ctx->horizon, ctx->params.dt_con, base, cmd, dyn, base_traj,
                                                   static_cast<hipStream_t>(stream)));
    return MPC_OK;
}

int mpc_prepare_tick(mpc_ctx *ctx, int count, const double *q, const double *dq, const double *ee_pose,
                     const double *base, const double *dyn_host, const double *prev_solution,
                     const uint8_t *has_solution, double *state0, double *base_traj, double *Jc, double *u_init,
                     double *x0, uint8_t *active, void *stream)
{
    int rc = check_ready(ctx);
    if (rc) return rc;
    if (count < 0 || !dyn_host) return fail(ctx, MPC_ERR_INVALID_ARG, "bad count / NULL dyn");
    for (int i = 0; i < 6; ++i)
        if (!std::isfinite(dyn_host[i]) || (i != 2 && i != 3 && !(dyn_host[i] > 0.0)))
            return fail(ctx, MPC_ERR_INVALID_ARG, "base dynamics: mass, I_z and the lags must be positive");
    if (count == 0) return MPC_OK;
    if (!q || !ee_pose || !base || !state0 || !base_traj || !Jc || !u_init || !active)
        return fail(ctx, MPC_ERR_INVALID_ARG, "q/ee_pose/base/state0/base_traj/Jc/u_init/active must be non-NULL");
    if ((prev_solution != nullptr) != (has_solution != nullptr))
        return fail(ctx, MPC_ERR_INVALID_ARG, "prev_solution and has_solution go together");
    if (x0 && prev_solution) {     // x0[i] = prev_solution[i + 6]: an in-place shift would read entries already overwritten
        const size_t len = (size_t)count * 6 * (size_t)ctx->horizon;
        if (x0 < prev_solution + len && prev_solution < x0 + len)
            return fail(ctx, MPC_ERR_INVALID_ARG, "x0 must not overlap prev_solution");
    }
    double dyn[6];
    std::memcpy(dyn, dyn_host, sizeof(dyn));
    MPC_HIP(ctx, hipSetDevice(ctx->device));
    MPC_HIP(ctx, mpck::launch_prepare_tick(ctx->params, count, dyn, q, dq, ee_pose, base, prev_solution, has_solution,
                                           state0, base_traj, Jc, u_init, x0, active,
                                           static_cast<hipStream_t>(stream)));
    return MPC_OK;
}

namespace {

int qp_workspace(mpc_ctx *ctx)
{
    if (ctx->d_qp_work) return MPC_OK;
    // as many teams as stay resident (LDS decides: two per CU at H = 20); each owns ~1 MB of workspace at H = 20
    hipDeviceProp_t prop;
    MPC_HIP(ctx, hipGetDeviceProperties(&prop, ctx->device));
    int teams = mpck::qp_teams_per_cu(ctx->horizon) * prop.multiProcessorCount;
    if (teams > ctx->max_scenarios) teams = ctx->max_scenarios;
    const size_t wd = mpck::qp_work_doubles(ctx->horizon), wi = mpck::qp_work_ints(ctx->horizon);
    while (teams > 1 && (size_t)teams * wd * sizeof(double) > (size_t(4) << 30)) teams /= 2;
    MPC_HIP(ctx, hipMalloc(&ctx->d_qp_work, (size_t)teams * wd * sizeof(double)));
    hipError_t e = hipMalloc(&ctx->d_qp_iwork, (size_t)teams * wi * sizeof(int));
    if (e != hipSuccess) {
        (void)hipFree(ctx->d_qp_work);
        ctx->d_qp_work = nullptr;
        return hip_fail(ctx, e, "hipMalloc(qp iwork)");
    }
    ctx->qp_teams = teams;
    return MPC_OK;
}

}  // namespace

int mpc_qp_solve(mpc_ctx *ctx, int count, const double *l, const double *g, const double *c, const double *nz,
                 const double *lo, const double *hi, double *s, double *y, int32_t *mode, void *stream)
{
    int rc = check_ready(ctx);
    if (rc) return rc;
    if (count < 0 || count > ctx->max_scenarios)
        return fail(ctx, MPC_ERR_INVALID_ARG, "count %d outside [0, %d]", count, ctx->max_scenarios);
    if (count == 0) return MPC_OK;
    if (!l || !g || !c || !nz || !lo || !hi || !s || !y || !mode)
        return fail(ctx, MPC_ERR_INVALID_ARG, "l/g/c/nz/lo/hi/s/y/mode must be non-NULL");
    MPC_HIP(ctx, hipSetDevice(ctx->device));
    rc = qp_workspace(ctx);
    if (rc) return rc;
    const int teams = count < ctx->qp_teams ? count : ctx->qp_teams;
    MPC_HIP(ctx, mpck::launch_qp(ctx->horizon, count, teams, l, g, c, nz, lo, hi, s, y, mode, ctx->d_qp_work, ctx->d_qp_iwork,
                                 static_cast<hipStream_t>(stream)));
    return MPC_OK;
}

int mpc_qp_step(mpc_ctx *ctx, int count, int rows, const int32_t *index, const int32_t *flags, const double *lo,
                const double *hi, const double *s_in, double *L, const double *G, const double *C, const double *NZ, double *R,
                double *V, double *s_out, double *r_out, int32_t *mode, void *stream)
{
    int rc = check_ready(ctx);
    if (rc) return rc;
    if (count < 0 || count > ctx->max_scenarios || rows < count)
        return fail(ctx, MPC_ERR_INVALID_ARG, "count %d outside [0, %d] or more requests than rows (%d)", count,
                    ctx->max_scenarios, rows);
    if (count == 0) return MPC_OK;
    if (!index || !flags || !lo || !hi || !s_in || !L || !G || !C || !NZ || !R || !V || !s_out || !r_out || !mode)
        return fail(ctx, MPC_ERR_INVALID_ARG, "every array of mpc_qp_step must be non-NULL");
    MPC_HIP(ctx, hipSetDevice(ctx->device));
    rc = qp_workspace(ctx);
    if (rc) return rc;
    const int teams = count < ctx->qp_teams ? count : ctx->qp_teams;
    MPC_HIP(ctx, mpck::launch_qp_step(ctx->horizon, count, teams, rows, index, flags, lo, hi, s_in, L, G, C, NZ, R, V, s_out, r_out,
                                      mode, ctx->d_qp_work, ctx->d_qp_iwork, static_cast<hipStream_t>(stream)));
    return MPC_OK;
}

size_t mpc_solve_workspace_bytes(int horizon, int rows)
{
    if (horizon <= 0 || horizon > MPC_MAX_HORIZON || rows <= 0) return 0;
    return mpck::solve_workspace_bytes(horizon, rows);
}

namespace {

int check_solve(mpc_ctx *ctx, int count, const void *ws, int rows, size_t ws_bytes)
{
    int rc = check_ready(ctx);
    if (rc) return rc;
    if (resolve_algo(ctx) != MPC_ALGO_PREFIX || !mpck::prefix_nz_supported(ctx->params))
        return fail(ctx, MPC_ERR_UNSUPPORTED, "device-resident solves need MPC_ALGO_PREFIX (UR twist pattern, H <= 170)");
    if (!ws || rows <= 0 || count < 0 || count > rows)
        return fail(ctx, MPC_ERR_INVALID_ARG, "workspace is NULL, or count %d outside [0, rows = %d]", count, rows);
    if ((reinterpret_cast<uintptr_t>(ws) & 255u) != 0) return fail(ctx, MPC_ERR_INVALID_ARG, "workspace must be 256-byte aligned");
    const size_t need = mpck::solve_workspace_bytes(ctx->horizon, rows);
    if (ws_bytes < need)
        return fail(ctx, MPC_ERR_INVALID_ARG, "workspace of %zu bytes, %zu needed for %d rows", ws_bytes, need, rows);
    return MPC_OK;
}

}  // namespace

int mpc_solve_begin(mpc_ctx *ctx, const mpc_scenarios *sc, const double *x0, double acc, int maxiter, void *ws, int rows,
                    size_t ws_bytes, void *stream)
{
    if (!sc) return fail(ctx, MPC_ERR_INVALID_ARG, "scenarios is NULL");
    int rc = check_solve(ctx, sc->count, ws, rows, ws_bytes);
    if (rc) return rc;
    rc = check_scenarios(ctx, sc, x0);
    if (rc) return rc;
    if (maxiter < 1 || !(acc == acc)) return fail(ctx, MPC_ERR_INVALID_ARG, "maxiter must be positive and acc a number");
    if (sc->count == 0) return MPC_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    MPC_HIP(ctx, hipSetDevice(ctx->device));
    rc = qp_workspace(ctx);                    // allocated here, so that mpc_solve_rounds never allocates
    if (rc) return rc;
    const mpck::SolveWorkspace W = mpck::carve_solve_workspace(ws, ctx->horizon, rows);
    MPC_HIP(ctx, mpck::launch_solve_begin(ctx->params, W, sc->count, x0, acc, maxiter, st));
    // mode 0 on entry: f, c, grad f and dc/dx at the (clipped) start point for everyone (SLSQP:419-425)
    MPC_HIP(ctx, mpck::launch_prefix_nz(ctx->params, ctx->d_decay, *sc, W.XE, W.F, W.G, W.C, W.NZ, st));
    if (!ctx->aug_seen) {
        // one word of host memory the QP kernel raises when it meets Kraft's augmented problem (rare: an inconsistent
        // linearisation); from the next mpc_solve_rounds call on, the rounds also launch the kernel that solves those
        MPC_HIP(ctx, hipHostMalloc(reinterpret_cast<void **>(&ctx->aug_seen), sizeof(int32_t), hipHostMallocMapped));
    }
    *ctx->aug_seen = 0;
    if (!ctx->solve_side) {
        MPC_HIP(ctx, hipStreamCreateWithFlags(&ctx->solve_side, hipStreamNonBlocking));
        for (int h = 0; h < 2; ++h) {
            MPC_HIP(ctx, hipEventCreateWithFlags(&ctx->solve_forked[h], hipEventDisableTiming));
            MPC_HIP(ctx, hipEventCreateWithFlags(&ctx->solve_joined[h], hipEventDisableTiming));
        }
    }
    ctx->solve_ws = ws;
    ctx->solve_round = 0;
    return MPC_OK;
}

int mpc_solve_rounds(mpc_ctx *ctx, const mpc_scenarios *sc, void *ws, int rows, size_t ws_bytes, int rounds, int bound,
                     int32_t *unfinished, void *stream)
{
    if (!sc) return fail(ctx, MPC_ERR_INVALID_ARG, "scenarios is NULL");
    int rc = check_solve(ctx, sc->count, ws, rows, ws_bytes);
    if (rc) return rc;
    if (sc->count > 0 && (!sc->state0 || !sc->target || !sc->base_traj || !sc->Jc))
        return fail(ctx, MPC_ERR_INVALID_ARG, "state0/target/base_traj/Jc must be non-NULL");
    if (rounds < 0 || bound < 0) return fail(ctx, MPC_ERR_INVALID_ARG, "rounds and bound must not be negative");
    if (sc->count == 0 || rounds == 0) return MPC_OK;
    if (!ctx->d_qp_work || ctx->solve_ws != ws)
        return fail(ctx, MPC_ERR_INVALID_ARG, "mpc_solve_begin has not run on this context with this workspace");
    if (bound > sc->count) bound = sc->count;
    hipStream_t st = static_cast<hipStream_t>(stream);
    MPC_HIP(ctx, hipSetDevice(ctx->device));
    const mpck::SolveWorkspace W = mpck::carve_solve_workspace(ws, ctx->horizon, rows);
    // A launch that fails part-way leaves the device lists and this context's round count out of step: the workspace
    // is then forgotten, and further calls are refused until mpc_solve_begin has run again.
#define MPC_SOLVE_HIP(call)                                      \
    do {                                                         \
        hipError_t e_ = (call);                                  \
        if (e_ != hipSuccess) {                                  \
            ctx->solve_ws = nullptr;                             \
            if (ctx->solve_side) (void)hipStreamSynchronize(ctx->solve_side);   /* nothing of ours is left running */ \
            return hip_fail(ctx, e_, #call);                     \
        }                                                        \
    } while (0)
    // has any round so far met an augmented QP request?  (Read without synchronisation: the word only ever goes from 0
    // to 1 during a solve, and a request that meets a round without the augmented kernel simply asks again.)
    const bool augmented = *static_cast<volatile int32_t *>(ctx->aug_seen) != 0;
    for (int r = 0; r < rounds && bound > 0; ++r) {     // bound == 0: nothing is unfinished, the state stays as it is
        const int round = ctx->solve_round, cur = (round + 1) & 1, cs = (round + 1) % 3;
        MPC_SOLVE_HIP(mpck::launch_solve_body(ctx->horizon, W, bound, round, st));
        ctx->solve_round = round + 1;
        mpck::RowList lf, lg, lq;
        lf.list = W.LIST_F[cur]; lf.count = W.counters + 8 * cs + 0; lf.bound = bound;
        lg.list = W.LIST_G[cur]; lg.count = W.counters + 8 * cs + 1; lg.bound = bound;
        lq.list = W.LIST_Q[cur]; lq.count = W.counters + 8 * cs + 2; lq.bound = bound;
        lq.head = W.counters + 8 * cs + 4;
        // The three kinds of request of a round belong to different problems: the evaluations run on a stream of their
        // own beside the QPs (the QP kernel's teams are persistent and fill every CU's LDS, so the evaluations' workgroups
        // find room as teams run out of requests: they take the place of the QP kernel's tail instead of following it;
        // measured 0.7 % on 65,536 problems, 1-2.5 % on 256 .. 32,768).  A handful of problems gains nothing from it and
        // pays the two stream dependencies (one problem: 14.2 -> 16.5 ms per solve), so they stay on one stream.
        const bool beside = MPC_SOLVE_OVERLAP && bound >= kSolveOverlapFrom;
        hipStream_t ev = beside ? ctx->solve_side : st;
        if (beside) {
            MPC_SOLVE_HIP(hipEventRecord(ctx->solve_forked[round & 1], st));
            MPC_SOLVE_HIP(hipStreamWaitEvent(ev, ctx->solve_forked[round & 1], 0));
        }
        // f and c where the line search asked for them (SLSQP:434-436)
        MPC_SOLVE_HIP(mpck::launch_prefix_rows(ctx->params, ctx->d_decay, *sc, W.XE, W.F, nullptr, W.C, nullptr, lf, ev));
        // grad f and dc/dx at the accepted point (SLSQP:438-440); f and c there are already in place
        MPC_SOLVE_HIP(mpck::launch_prefix_rows(ctx->params, ctx->d_decay, *sc, W.XE, W.F, W.G, nullptr, W.NZ, lg, ev));
        if (beside) MPC_SOLVE_HIP(hipEventRecord(ctx->solve_joined[round & 1], ev));
        // BFGS update + QP of the problems that start a major iteration
        MPC_SOLVE_HIP(mpck::launch_qp_step_rows(ctx->horizon, ctx->qp_teams, rows, lq, W.QPFLAGS, W.LO, W.HI, W.S, W.L, W.G, W.C,
                                                W.NZ, W.RM, W.V, W.QPMODE, ctx->d_qp_work, ctx->d_qp_iwork, ctx->aug_seen, false,
                                                W.counters + mpck::kSolveFreeSteps, st));
        if (augmented) {                                    // ... and Kraft's augmented problems, where there are any
            lq.head = W.counters + 8 * cs + 5;
            MPC_SOLVE_HIP(mpck::launch_qp_step_rows(ctx->horizon, ctx->qp_teams, rows, lq, W.QPFLAGS, W.LO, W.HI, W.S, W.L, W.G,
                                                    W.C, W.NZ, W.RM, W.V, W.QPMODE, ctx->d_qp_work, ctx->d_qp_iwork, nullptr, true,
                                                    nullptr, st));
        }
        if (beside) MPC_SOLVE_HIP(hipStreamWaitEvent(st, ctx->solve_joined[round & 1], 0));
    }
    const int slot = ctx->solve_round % 3;
    MPC_SOLVE_HIP(mpck::launch_solve_count(W, slot, st));
    if (unfinished)
        MPC_SOLVE_HIP(hipMemcpyAsync(unfinished, W.counters + 8 * slot + 3, sizeof(int32_t), hipMemcpyDefault, st));
#undef MPC_SOLVE_HIP
    return MPC_OK;
}

int mpc_solve_run(mpc_ctx *ctx, const mpc_scenarios *sc, void *ws, int rows, size_t ws_bytes, int maxiter, int32_t *unfinished,
                  void *stream)
{
    if (!sc) return fail(ctx, MPC_ERR_INVALID_ARG, "scenarios is NULL");
    int rc = check_solve(ctx, sc->count, ws, rows, ws_bytes);
    if (rc) return rc;
    if (sc->count > 0 && (!sc->state0 || !sc->target || !sc->base_traj || !sc->Jc))
        return fail(ctx, MPC_ERR_INVALID_ARG, "state0/target/base_traj/Jc must be non-NULL");
    if (maxiter < 1) return fail(ctx, MPC_ERR_INVALID_ARG, "maxiter must be positive");
    if (!mpck::solve_team_supported(ctx->params))
        return fail(ctx, MPC_ERR_UNSUPPORTED, "mpc_solve_run needs the QP's vectors in LDS and one wavefront per evaluation (H <= 21)");
    if (sc->count == 0) return MPC_OK;
    if (!ctx->d_qp_work || ctx->solve_ws != ws || ctx->solve_round != 0)
        return fail(ctx, MPC_ERR_INVALID_ARG, "mpc_solve_run follows mpc_solve_begin on this context and workspace, before any round");
    hipStream_t st = static_cast<hipStream_t>(stream);
    MPC_HIP(ctx, hipSetDevice(ctx->device));
    const mpck::SolveWorkspace W = mpck::carve_solve_workspace(ws, ctx->horizon, rows);
    hipError_t e = mpck::launch_solve_team(ctx->params, ctx->d_decay, *sc, W, sc->count, maxiter, ctx->qp_teams, ctx->d_qp_work,
                                           ctx->d_qp_iwork, st);
    if (e == hipSuccess && unfinished) e = hipMemcpyAsync(unfinished, W.counters + 3, sizeof(int32_t), hipMemcpyDefault, st);
    if (e != hipSuccess) {
        ctx->solve_ws = nullptr;               // the state is as far as the launches got: begin again
        return hip_fail(ctx, e, "mpc_solve_run");
    }
    return MPC_OK;
}

int mpc_solve_result(mpc_ctx *ctx, int count, void *ws, int rows, size_t ws_bytes, double *x, double *fun, int32_t *nit,
                     int32_t *status, int32_t *stats, void *stream)
{
    int rc = check_solve(ctx, count, ws, rows, ws_bytes);
    if (rc) return rc;
    if (count == 0) return MPC_OK;
    if (!x || !fun || !nit || !status) return fail(ctx, MPC_ERR_INVALID_ARG, "x/fun/nit/status must be non-NULL");
    if (ctx->solve_ws != ws) return fail(ctx, MPC_ERR_INVALID_ARG, "mpc_solve_begin has not run on this context with this workspace");
    MPC_HIP(ctx, hipSetDevice(ctx->device));
    const mpck::SolveWorkspace W = mpck::carve_solve_workspace(ws, ctx->horizon, rows);
    MPC_HIP(ctx, mpck::launch_solve_result(ctx->horizon, W, count, ctx->solve_round % 3, x, fun, nit, status, stats,
                                           static_cast<hipStream_t>(stream)));
    return MPC_OK;
}

int mpc_gather_scenarios(mpc_ctx *ctx, const mpc_scenarios *src, const int32_t *index, int count,
                         const mpc_scenarios *dst, void *stream)
{
    int rc = check_ready(ctx);
    if (rc) return rc;
    if (!src || !dst || count < 0 || src->count < 0) return fail(ctx, MPC_ERR_INVALID_ARG, "bad src/dst/count");
    if (count == 0) return MPC_OK;
    if (!index) return fail(ctx, MPC_ERR_INVALID_ARG, "index is NULL");
    if (!src->state0 || !src->target || !src->base_traj || !src->Jc || !dst->state0 || !dst->target ||
        !dst->base_traj || !dst->Jc)
        return fail(ctx, MPC_ERR_INVALID_ARG, "state0/target/base_traj/Jc must be non-NULL on both sides");
    if ((src->has_prev && !src->u_prev) || (dst->has_prev && !dst->u_prev))
        return fail(ctx, MPC_ERR_INVALID_ARG, "has_prev given without u_prev");
    if ((src->u_prev != nullptr) != (dst->u_prev != nullptr) || (src->has_prev != nullptr) != (dst->has_prev != nullptr))
        return fail(ctx, MPC_ERR_INVALID_ARG, "u_prev/has_prev must be present on both sides or on neither");
    MPC_HIP(ctx, hipSetDevice(ctx->device));
    MPC_HIP(ctx, mpck::launch_gather_scenarios(ctx->horizon, *src, index, count, *dst,
                                               static_cast<hipStream_t>(stream)));
    return MPC_OK;
}

}  // extern "C"
