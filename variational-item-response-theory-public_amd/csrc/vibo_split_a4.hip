// row-split ELBO kernel, template ability width 4 (see vibo_split_kernel.hpp)
#include "vibo_split_kernel.hpp"
#include "vibo_launch.hpp"
namespace vibo {
hipError_t launch_elbo_split_a4(const ElboParams& p, int irt, bool grad, int nq, int grid, hipStream_t s) {
    return launch_split_at<4>(p, irt, grad, nq, grid, s);
}
}  // namespace vibo
