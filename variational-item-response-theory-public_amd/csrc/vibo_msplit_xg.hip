// matrix row-split ELBO kernel, conditional posterior with the experts' sums formed in the kernel (XM == 3), fp32 rows through row_index
#include "vibo_msplit_kernel.hpp"
#include "vibo_launch.hpp"
namespace vibo {
hipError_t launch_elbo_msplit_xg(const ElboParams& p, int irt, bool grad, int nw, int grid, hipStream_t s) {
    return launch_msplit_fused<1>(p, irt, grad, nw, grid, s);
}
}  // namespace vibo
