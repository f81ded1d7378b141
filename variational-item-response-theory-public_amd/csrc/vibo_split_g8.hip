// row-split ELBO kernel gathering fp32 rows through row_index (shuffled minibatches), template ability width 8
#include "vibo_split_kernel.hpp"
#include "vibo_launch.hpp"
namespace vibo {
hipError_t launch_elbo_split_g8(const ElboParams& p, int irt, bool grad, int nq, int grid, hipStream_t s) {
    return launch_split_at<8, 1>(p, irt, grad, nq, grid, s);
}
}  // namespace vibo
