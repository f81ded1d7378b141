// multi-sample forward kernel instantiations (see vibo_multi_kernel.hpp)
#include "vibo_multi_kernel.hpp"
#include "vibo_multi.hpp"
namespace vibo {
hipError_t launch_elbo_multi(const MultiParams& mp, int at, int irt, int sc, int nq, int grid, hipStream_t s) {
    if (at <= 2) return launch_multi_at<2>(mp, irt, sc, nq, grid, s);
    if (at == 4) return launch_multi_at<4>(mp, irt, sc, nq, grid, s);
    return launch_multi_at<8>(mp, irt, sc, nq, grid, s);
}
}  // namespace vibo
