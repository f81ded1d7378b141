// row-split ELBO kernel, template ability width 8 (see vibo_split_kernel.hpp)
#include "vibo_split_kernel.hpp"
#include "vibo_launch.hpp"
namespace vibo {
hipError_t launch_elbo_split_a8(const ElboParams& p, int irt, bool grad, int nq, int grid, hipStream_t s) {
    return launch_split_at<8>(p, irt, grad, nq, grid, s);
}
}  // namespace vibo
