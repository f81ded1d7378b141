// row-split ELBO kernel, template ability width 2 (see vibo_split_kernel.hpp)
#include "vibo_split_kernel.hpp"
#include "vibo_launch.hpp"
namespace vibo {
hipError_t launch_elbo_split_a2(const ElboParams& p, int irt, bool grad, int nq, int grid, hipStream_t s) {
    return launch_split_at<2>(p, irt, grad, nq, grid, s);
}
}  // namespace vibo
