// matrix row-split ELBO kernel, 1-byte cell codes (VIBO_MASK_CODES) (see vibo_msplit_kernel.hpp)
#include "vibo_msplit_kernel.hpp"
#include "vibo_launch.hpp"
namespace vibo {
hipError_t launch_elbo_msplit_c(const ElboParams& p, int irt, bool grad, int nw, int grid, hipStream_t s) {
    return launch_msplit_rm<2, false>(p, irt, grad, nw, grid, s);
}
}  // namespace vibo
