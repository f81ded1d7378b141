// matrix row-split ELBO kernel, fp32 rows gathered through row_index (see vibo_msplit_kernel.hpp)
#include "vibo_msplit_kernel.hpp"
#include "vibo_launch.hpp"
namespace vibo {
hipError_t launch_elbo_msplit_g(const ElboParams& p, int irt, bool grad, int nw, int grid, hipStream_t s) {
    return launch_msplit_rm<1, false>(p, irt, grad, nw, grid, s);
}
}  // namespace vibo
