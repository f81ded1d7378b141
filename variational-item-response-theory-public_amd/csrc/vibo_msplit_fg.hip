// matrix row-split ELBO kernel with planar flows, fp32 rows gathered through row_index (see vibo_msplit_kernel.hpp)
#include "vibo_msplit_kernel.hpp"
#include "vibo_launch.hpp"
namespace vibo {
hipError_t launch_elbo_msplit_fg(const ElboParams& p, int irt, bool grad, int nw, int grid, hipStream_t s) {
    return launch_msplit_rm<1, true>(p, irt, grad, nw, grid, s);
}
}  // namespace vibo
