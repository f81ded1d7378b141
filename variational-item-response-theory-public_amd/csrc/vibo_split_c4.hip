// row-split ELBO kernel reading 1-byte cell codes (VIBO_MASK_CODES), template ability width 4
#include "vibo_split_kernel.hpp"
#include "vibo_launch.hpp"
namespace vibo {
hipError_t launch_elbo_split_c4(const ElboParams& p, int irt, bool grad, int nq, int grid, hipStream_t s) {
    return launch_split_at<4, 2>(p, irt, grad, nq, grid, s);
}
}  // namespace vibo
