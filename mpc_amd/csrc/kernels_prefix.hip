// MPC_ALGO_PREFIX — placeholder until the shared-prefix kernel lands.
#include "mpc_kernels.hpp"

namespace mpck {

bool prefix_supported(const mpc_params &) { return false; }

hipError_t launch_prefix(const mpc_params &, const double *, const mpc_scenarios &, const double *, double *,
                         double *, double *, double *, hipStream_t)
{
    return hipErrorNotSupported;
}

}  // namespace mpck
