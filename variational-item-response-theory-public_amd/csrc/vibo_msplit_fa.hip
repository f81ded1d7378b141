// matrix row-split ELBO kernel with planar flows, fp32 rows in order (see vibo_msplit_kernel.hpp)
#include "vibo_msplit_kernel.hpp"
#include "vibo_launch.hpp"
namespace vibo {
hipError_t launch_elbo_msplit_fa(const ElboParams& p, int irt, bool grad, int nw, int grid, hipStream_t s) {
    return launch_msplit_rm<0, true>(p, irt, grad, nw, grid, s);
}
}  // namespace vibo
